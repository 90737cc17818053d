"""Alternating GPU samplers (rlpyt/samplers/parallel/gpu/alternating_sampler.py:6-83 with the action
servers of rlpyt/samplers/parallel/gpu/action_server.py:123-363) on the HBM-resident sampler.

The reference forks two sets of env workers that share CPUs in pairs; while one set steps its half of
the environments the action server computes the actions of the other half.  ``GpuSampler``'s pipeline
groups are the same idea with N halves (``samplers/gpu.py``); the alternating samplers are the
reference's two-set contract on top of it:

* ``batch_B`` even, the environments split into two halves (columns ``[:B/2]`` and ``[B/2:]`` of the
  batch, as ``_make_alternating_pairs``), an even number of worker processes taken from
  ``affinity["workers_cpus"]`` (which must be an ``alternating`` affinity when one is given), the first
  half of the workers stepping the first half of the environments -- ``split_workers`` with two groups;
* feed-forward agents get ``agent.alternating = True``; a recurrent agent must declare itself
  ``alternating`` (the reference's ``AlternatingRecurrentAgentMixin`` keeps one RNN state per half and
  toggles; here every pipeline group owns its state in the device batch, ``samplers/device.py``);
* ``AlternatingSampler``: the halves overlap freely -- a half steps as soon as ITS actions are there
  (``AlternatingActionServer``); all of ``GpuSampler``'s machinery applies (captured step graphs, native
  serve loop);
* ``NoOverlapAlternatingSampler``: a half is released only once the other half has finished stepping
  (``NoOverlapAlternatingActionServer.serve_actions``), so paired workers never compete for their shared
  CPU; the hand-off order is the reference's, statement for statement, in ``_serve_batch`` /
  ``_tail_batch`` below (Python loop: the native loop releases a group as soon as its actions exist).

Batches have the same semantics as ``GpuSampler``'s: every column is one environment's trajectory under
one fixed policy (tests/test_alternating.py holds both samplers to ``GpuSampler`` batches field for field).
"""
import time

from .gpu import GpuSampler


class AlternatingSamplerBase(GpuSampler):
    alternating = True

    def __init__(self, *args, **kwargs):
        kwargs["n_groups"] = 2
        kwargs["split_workers"] = True
        super().__init__(*args, **kwargs)
        assert self.batch_spec.B % 2 == 0, "Need even number for sampler batch_B."

    def _split_min_workers(self):
        return 2            # one worker per half is enough (the reference: n_worker // 2 each)

    def _resolve_layout(self, affinity):
        super()._resolve_layout(affinity)
        self.n_groups = 2 if self.batch_spec.B >= 2 else 1
        if affinity is not None and affinity.get("workers_cpus", None) is not None \
                and self._n_workers_arg is None:
            assert affinity.get("alternating", False), "Need alternating affinity."
        if self.n_workers > 0:
            assert self.n_workers % 2 == 0, "Need even number workers."

    def initialize(self, agent, *args, **kwargs):
        if agent.recurrent and not agent.alternating:
            raise TypeError("If agent is recurrent, must be 'alternating' to use here.")
        elif not agent.recurrent:
            agent.alternating = True   # FF agent doesn't need a special class, but tell it so
        if self.eval_n_envs:
            assert self.eval_n_envs % 2 == 0
        examples = super().initialize(agent, *args, **kwargs)
        self.half_B = self.batch_spec.B // 2
        assert [G.Bg for G in self.groups] == [self.half_B] * 2
        return examples


class AlternatingSampler(AlternatingSamplerBase):
    """Two halves, free overlap (``AlternatingActionServer``)."""


class NoOverlapAlternatingSampler(AlternatingSamplerBase):
    """Two halves that never step at the same time (``NoOverlapAlternatingActionServer``)."""

    def __init__(self, *args, **kwargs):
        kwargs["native_loop"] = False
        super().__init__(*args, **kwargs)

    def _serve_batch(self, dev, par, T, tm, completed):
        if not par:             # envs stepped inline by the master: nothing can overlap anyway
            return super()._serve_batch(dev, par, T, tm, completed)
        sync, groups = self.sync, dev.groups
        for t in range(T):
            for alt in range(2):
                G = groups[alt]
                t0 = time.perf_counter()
                sync.master_wait_obs(G.idx)          # this half wrote obs / reward (and stopped stepping)
                tm["wait_env_s"] += time.perf_counter() - t0
                if t > 0 or alt > 0:                 # only now may the OTHER half go
                    dev.finish(groups[1 - alt])
                    sync.master_post_act(groups[1 - alt].idx)
                dev.issue(G, t, first=(t == 0))
        return False

    def _tail_batch(self, dev, par):
        if not par:
            return super()._tail_batch(dev, par)
        sync, groups = self.sync, dev.groups
        for alt in range(2):
            G = groups[alt]
            sync.master_wait_obs(G.idx)
            if alt == 0:                             # the second half's last step
                dev.finish(groups[1])
                sync.master_post_act(groups[1].idx)
            dev.tail(G)
