"""What the replay-based algorithms of this path (DQN, CategoricalDQN, R2D1) share, as data and
small objects instead of a method chain:

* ``UpdatePlan`` -- everything that follows from (sampler batch size, training batch size, replay
  ratio, the step-denominated thresholds of the constructor): updates per iteration, the first
  learning iteration, the epsilon-greedy schedule's iteration range, the length of the importance
  exponent's anneal;
* ``BetaAnneal`` -- the prioritized-replay importance exponent as a function of the iteration;
* ``ReplayFeed`` -- which leaves of a sampler batch (or of the runner's ``examples`` dict) become
  which field of a replay record, as a table;
* ``UpdateLog`` -- per-update diagnostics kept on the device until the end of ``optimize_agent``
  (ONE device-to-host copy per call instead of one ``.item()`` per update and field).

The hyper-parameter semantics are the reference's (rlpyt/algos/dqn/dqn.py:76-132,267-279,
rlpyt/algos/dqn/r2d1.py:84-130) -- they have to be, the runs are compared update for update
(tests/test_algo_parity.py)."""
from collections import namedtuple

import torch

from ...utils.collections import namedarraytuple
from ...utils.deferred import PendingOptInfo

UpdatePlan = namedtuple("UpdatePlan", ["updates_per_itr", "first_learn_itr", "eps_last_itr",
                                       "beta_last_itr"])


def plan_updates(sampler_batch, train_batch, replay_ratio, min_steps_learn, eps_steps,
                 pri_beta_steps):
    """``replay_ratio`` = consumed / generated samples, so one iteration of ``sampler_batch`` new
    steps pays for ``ratio * sampler_batch / train_batch`` updates (at least one)."""
    return UpdatePlan(
        updates_per_itr=max(1, round(replay_ratio * sampler_batch / train_batch)),
        first_learn_itr=int(min_steps_learn // sampler_batch),
        eps_last_itr=max(1, int(eps_steps // sampler_batch)),
        beta_last_itr=max(1, pri_beta_steps // sampler_batch))


class BetaAnneal:
    """Linear from ``start`` at the first learning iteration to ``final`` at ``last_itr``; None
    (= leave the buffer alone) once past ``last_itr``."""

    def __init__(self, start, final, first_itr, last_itr):
        self.start, self.final, self.first_itr, self.last_itr = start, final, first_itr, last_itr

    def at(self, itr):
        if itr > self.last_itr:
            return None
        frac = min(1, max(0, itr - self.first_itr) / (self.last_itr - self.first_itr))
        return frac * self.final + (1 - frac) * self.start


class ReplayFeed:
    """Sampler batch -> replay record.  ``routes``: (record field, path into the Samples tree);
    the runner's ``examples`` dict is addressed by the last path component (``agent_info`` leaves
    by ``("agent_info", leaf)``)."""

    STEP = (("observation", ("env", "observation")), ("action", ("agent", "action")),
            ("reward", ("env", "reward")), ("done", ("env", "done")))
    RNN = STEP + (("prev_rnn_state", ("agent", "agent_info", "prev_rnn_state")),)

    def __init__(self, routes, name):
        self.routes = routes
        self.Record = namedarraytuple(name, [field for field, _ in routes])

    @staticmethod
    def _walk(root, path):
        for step in path:
            root = root[step] if isinstance(root, dict) else getattr(root, step)
        return root

    def from_samples(self, samples):
        return self.Record(*(self._walk(samples, path) for _, path in self.routes))

    def from_examples(self, examples):
        return self.Record(*(self._walk(examples, path[1:] if path[0] in ("env", "agent") else path)
                             for _, path in self.routes))


Prioritised = namedarraytuple("PrioritiesSamplesToBuffer", ["priorities", "samples"])


class UpdateLog:
    """Accumulates one row of scalar diagnostics and any number of vector diagnostics per update,
    all as device tensors; ``to_opt_info`` starts ONE device -> host copy of them and returns an
    ``OptInfo`` that waits for it on first access (``utils/deferred.py``)."""

    def __init__(self, OptInfo, scalar_fields):
        self.OptInfo, self.scalar_fields = OptInfo, tuple(scalar_fields)
        self.blocks, self.vectors = [], {f: [] for f in OptInfo._fields if f not in scalar_fields}

    def add(self, scalars, **vectors):
        self.blocks.append(torch.stack([s.detach().float() for s in scalars]).unsqueeze(0))
        for name, v in vectors.items():
            self.vectors[name].append(v.reshape(-1))

    def add_rows(self, rows, **vectors):
        """Several updates at once: ``rows [k, n_scalars]`` and ``[k, m]`` vector diagnostics, all
        device tensors (the rings a captured update graph writes; cloned: the rings are reused)."""
        self.blocks.append(rows.detach().float().clone())
        for name, v in vectors.items():
            self.vectors[name].append(v.detach().clone().reshape(-1))

    def to_opt_info(self):
        fields = self.OptInfo._fields
        if not self.blocks:
            return self.OptInfo(*([] for _ in fields))
        one = lambda parts: parts[0] if len(parts) == 1 else torch.cat(parts)    # noqa: E731
        names = [f for f, chunks in self.vectors.items() if chunks]
        tensors = [one(self.blocks)] + [one(self.vectors[f]) for f in names]
        OptInfo, scalar_fields = self.OptInfo, self.scalar_fields

        def build(host):
            out = {f: [] for f in fields}
            table = host[0].tolist()
            for k, f in enumerate(scalar_fields):
                out[f] = [r[k] for r in table]
            for f, h in zip(names, host[1:]):
                out[f] = h.tolist()
            return OptInfo(*(out[f] for f in fields))

        return PendingOptInfo(OptInfo, tensors, build)
