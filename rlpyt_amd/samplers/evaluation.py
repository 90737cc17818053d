"""Offline evaluation of the MI355X rollout sampler: separate eval envs stepped by the same worker
processes, the batched forward on the device -- the role of ``ParallelSamplerBase.evaluate_agent`` +
``ActionServer.serve_actions_evaluation`` + the eval collectors (rlpyt/samplers/parallel/base.py:
88-98,115-145, gpu/action_server.py:76-120, gpu/collectors.py:129-161)."""
import queue as queue_mod

import numpy as np
import torch

from ..utils import logger
from ..utils.buffer import _map, buffer_from_example
from ..utils.collections import AttrDict
from .collections import StepBuffer
from .device import copy_leaves
from .workers import EVAL_TRAJ_CHECK, EvalRunner


class Evaluator:
    def __init__(self, sampler, obs_example, action_example, n_w, shared):
        """``eval_n_envs`` is spread evenly over the workers (at least one each)."""
        s = self.s = sampler
        if not s.eval_max_steps:
            raise ValueError("GpuSampler: eval_n_envs > 0 needs eval_max_steps (total env steps "
                             "of one evaluation), as in the reference's samplers.")
        per = max(1, s.eval_n_envs // n_w)
        Be = per * n_w
        if Be != s.eval_n_envs:
            logger.log(f"GpuSampler: using {Be} evaluation environments ({per} per worker).")
        s.eval_n_envs = Be
        s.eval_max_T = self.max_T = max_T = max(1, int(s.eval_max_steps // Be))
        kwargs = s.eval_env_kwargs if s.eval_env_kwargs is not None else s.env_kwargs
        envs = [s.EnvCls(**kwargs) for _ in range(Be)]
        for i, env in enumerate(envs):
            env.seed(s.seed + 50000 + s.rank * Be + i)
        self.Be = Be
        self.step_np = StepBuffer(
            observation=buffer_from_example(obs_example, (Be,), share_memory=shared),
            action=buffer_from_example(action_example, (Be,), share_memory=shared),
            reward=buffer_from_example(np.asarray(0, dtype=np.float32), (Be,), share_memory=shared),
            done=buffer_from_example(np.asarray(False), (Be,), share_memory=shared))
        self.runners = [EvalRunner(envs[w * per:(w + 1) * per], self.step_np[w * per:(w + 1) * per],
                                   s.TrajInfoCls, max_T) for w in range(n_w)]
        self.dev = None

    def _ensure_device(self):
        if self.dev is not None:
            return self.dev
        from ..utils.buffer import torchify_buffer
        s, Be = self.s, self.Be
        dev = s.agent.device
        self.step_pyt = torchify_buffer(self.step_np)
        self.dev = AttrDict(
            obs=buffer_from_example(s.examples["observation"], (Be,), device=dev),
            act=buffer_from_example(s.examples["action"], (Be,), device=dev),
            rew=torch.zeros(Be, dtype=torch.float32, device=dev),
            done=torch.zeros(Be, dtype=torch.bool, device=dev))
        return self.dev

    def run(self, itr):
        """Evaluate with the agent's current parameters (the caller has put the agent in eval
        mode).  Observations go up and actions come down once per step (eager launches --
        evaluation is outside the timed training path).  Stops after ``eval_max_steps`` env steps
        or, if given, once ``eval_max_trajectories`` have completed (checked every EVAL_TRAJ_CHECK
        steps).  Returns the completed TrajInfos."""
        s = self.s
        D = self._ensure_device()
        agent = s.agent
        par = s.n_workers > 0
        cuda = agent.device.type == "cuda"
        step_np, step_pyt = self.step_np, self.step_pyt
        traj_infos = []

        def take(info):
            ti = s.TrajInfoCls()
            ti.update(info)
            traj_infos.append(ti)

        def drain(block_for_sentinels=0):
            q, n_sent = s.ctrl.eval_traj_infos_queue, 0
            while True:
                try:
                    item = q.get(block=block_for_sentinels > 0, timeout=20)
                except queue_mod.Empty:
                    if block_for_sentinels > 0:
                        raise RuntimeError("GpuSampler.evaluate_agent: an env worker did not "
                                           "finish its evaluation run.")
                    return
                if item is None:
                    n_sent += 1
                    if n_sent >= block_for_sentinels > 0:
                        return
                else:
                    take(item)

        agent.reset()
        agent.select_envs(None, None)
        if agent.recurrent:
            agent.select_slot("eval")
        g_eval = len(s.groups)
        if par:
            s.ctrl.stop_eval.value = False
            s.ctrl.do_eval.value = True
            s.sync.master_start_batch()
        else:
            self.runners[0].begin()
        stop = False
        for t in range(self.max_T):
            if par:
                if t % EVAL_TRAJ_CHECK == 0:
                    drain()
                s.sync.master_wait_obs(g_eval)
            dn = step_np.done
            if np.any(dn):      # null prev action / reward after a reset (action_server.py:95-98)
                _map(lambda x: x.__setitem__(dn, 0), step_np.action)
                step_np.reward[dn] = 0
            copy_leaves(D.obs, step_pyt.observation, non_blocking=False)
            copy_leaves(D.act, step_pyt.action)
            D.rew.copy_(step_pyt.reward)
            if agent.recurrent:
                D.done.copy_(step_pyt.done)
                agent.reset_where(D.done)
            action, _agent_info = agent.step(D.obs, D.act, D.rew)
            copy_leaves(step_pyt.action, action)     # D2H (synchronous)
            if s.eval_max_trajectories is not None and t % EVAL_TRAJ_CHECK == 0:
                stop = len(traj_infos) >= s.eval_max_trajectories
            if par:
                s.ctrl.stop_eval.value = stop
                s.sync.master_post_act(g_eval)
            elif not stop:
                self.runners[0].step_all(take)
            if stop:
                logger.log(f"Evaluation reached max num trajectories "
                           f"({s.eval_max_trajectories}).")
                break
        if not stop and s.eval_max_trajectories is not None:
            logger.log(f"Evaluation reached max num time steps ({self.max_T}).")
        if par:
            s.sync.master_wait_obs(g_eval)     # the workers' closing arrival
            s.sync.master_wait_batch_done()
            drain(block_for_sentinels=s.n_workers)
            s.ctrl.do_eval.value = False
        if cuda:
            torch.cuda.current_stream().synchronize()
        return traj_infos
