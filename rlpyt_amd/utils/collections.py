"""``namedarraytuple``: the [Time, Batch, ...] container the whole reference API speaks.

Behavioural contract restated from rlpyt/utils/collections.py:16-133:
a namedtuple subclass whose ``x[loc]`` indexes EVERY field (recursing through nested
tuples, leaving ``None`` fields alone) and returns the same type; ``x[loc] = v`` assigns
field-by-field when ``v`` has the same fields, else broadcasts ``v`` into every non-None
field; ``in`` tests field names; ``get(i)`` is raw tuple indexing; ``items()`` yields
``(name, value)``.  Fields may be numpy arrays, CPU tensors or HBM tensors alike.
"""
import collections
import sys

RESERVED_NAMES = ("get", "items")


class AttrDict(dict):
    """dict whose keys are also attributes (used by TrajInfo)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self

    def copy(self):
        return type(self)(**{k: (v.copy() if isinstance(v, dict) else v)
                             for k, v in self.items()})


class _NamedArrayTupleBase:
    """Mixin carrying the array-style behaviour; combined with a namedtuple class."""
    __slots__ = ()

    def __getitem__(self, loc):
        out = []
        for name, field in zip(self._fields, self):
            if field is None:
                out.append(None)
                continue
            try:
                out.append(field[loc])
            except IndexError as e:
                raise Exception(f"Occured in {type(self)} at field '{name}'.") from e
        return type(self)(*out)

    def __setitem__(self, loc, value):
        same = isinstance(value, tuple) and getattr(value, "_fields", None) == self._fields
        for j, field in enumerate(self):
            v = tuple.__getitem__(value, j) if same else value
            if field is None and (v is None or not same):
                continue
            try:
                field[loc] = v
            except (ValueError, IndexError, TypeError) as e:
                raise Exception(
                    f"Occured in {type(self)} at field '{self._fields[j]}'.") from e

    def __contains__(self, key):
        return key in self._fields

    def get(self, index):
        return tuple.__getitem__(self, index)

    def items(self):
        return zip(self._fields, self)


def namedarraytuple(typename, field_names, return_namedtuple_cls=False,
                    classname_suffix=False):
    """Create a namedarraytuple class (see module docstring)."""
    nt_name = typename + "_nt" if classname_suffix else typename
    if classname_suffix:
        typename = typename + "_nat"
    try:
        module = sys._getframe(1).f_globals.get("__name__", "__main__")
    except (AttributeError, ValueError):
        module = None
    nt_cls = collections.namedtuple(nt_name, field_names, module=module)
    for name in nt_cls._fields:
        if name in RESERVED_NAMES:
            raise ValueError(f"Disallowed field name: {name}.")
    cls = type(typename, (_NamedArrayTupleBase, nt_cls), {"__slots__": ()})
    cls.__module__ = nt_cls.__module__
    cls.__doc__ = f"{typename}({', '.join(nt_cls._fields)})"
    if return_namedtuple_cls:
        return cls, nt_cls
    return cls


def is_namedtuple_class(obj):
    return isinstance(obj, type) and issubclass(obj, tuple) and hasattr(obj, "_fields")


def is_namedarraytuple(obj):
    return isinstance(obj, _NamedArrayTupleBase)


def is_namedtuple(obj):
    return isinstance(obj, tuple) and hasattr(obj, "_fields")


def namedarraytuple_like(example, classname_suffix=False):
    """A namedarraytuple class with the fields of ``example`` (instance or class)."""
    if is_namedarraytuple(example):
        return type(example)
    if is_namedtuple(example) or is_namedtuple_class(example):
        name = example.__name__ if isinstance(example, type) else type(example).__name__
        return namedarraytuple(name, example._fields, classname_suffix=classname_suffix)
    raise TypeError(f"expected a namedtuple instance or class, got {type(example)}")
