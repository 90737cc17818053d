"""``save__init__args``: store constructor arguments as attributes (the reference's
idiom for every component class, rlpyt/utils/quick_args.py:5-25)."""
import sys


def save__init__args(values, underscore=False, overwrite=False, subclass_only=False):
    """Call as ``save__init__args(locals())`` inside ``__init__``: every named parameter
    of that ``__init__`` (not ``*args``/``**kwargs``) becomes ``self.<name>``; attributes
    that already exist are kept unless ``overwrite``."""
    self = values["self"]
    code = sys._getframe(1).f_code
    names = code.co_varnames[:code.co_argcount + code.co_kwonlyargcount]
    prefix = "_" if underscore else ""
    for name in names:
        if name == "self" or name not in values:
            continue
        attr = prefix + name
        if overwrite or not hasattr(self, attr):
            setattr(self, attr, values[name])
