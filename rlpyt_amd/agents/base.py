"""Agent protocol (rlpyt/agents/base.py:17-245): the object the sampler steps and the
algorithm differentiates through.

MI355X-first difference from the reference: outputs stay in HBM.  The reference's
``__call__``/``step`` copy model outputs back to the host every call
(agents/pg/categorical.py:25,42) because its losses and collectors run on CPU tensors;
here the loss kernels and the rollout buffers live on the device, so outputs are returned
where the model computed them unless ``host_outputs=True`` (drop-in mode under the
reference's own CPU samplers/algos).
"""
import torch

from ..models.utils import strip_ddp_state_dict
from ..utils import logger
from ..utils.collections import namedarraytuple
from ..utils.deferred import LEAN_HOST

AgentInputs = namedarraytuple("AgentInputs", ["observation", "prev_action", "prev_reward"])
AgentStep = namedarraytuple("AgentStep", ["action", "agent_info"])
AgentInputsRnn = namedarraytuple("AgentInputsRnn",
                                 ["observation", "prev_action", "prev_reward", "init_rnn_state"])


class BaseAgent:
    recurrent = False
    alternating = False

    def __init__(self, ModelCls=None, model_kwargs=None, initial_model_state_dict=None,
                 host_outputs=False):
        self.ModelCls, self.model_kwargs = ModelCls, dict(model_kwargs or {})
        self.initial_model_state_dict, self.host_outputs = initial_model_state_dict, host_outputs
        self.model = self.shared_model = self.distribution = None
        self.device = torch.device("cpu")
        self._mode = None
        self._mode_itr = None
        self._mode_repeat = False
        self.sample_generator = None   # torch.Generator for action draws (None: default)
        # (uniform table [T', B], device row index) for agents whose sampling forward can draw
        # from pre-generated uniforms (keeps RNG state out of captured hipGraphs)
        self.sample_uniforms = None

    def __call__(self, observation, prev_action, prev_reward):
        raise NotImplementedError

    def _new_model(self, state_dict=None):
        """A model instance for this agent's env (``make_env_to_model_kwargs``) and settings,
        optionally loaded with ``state_dict``."""
        model = self.ModelCls(**self.env_model_kwargs, **self.model_kwargs)
        if state_dict is not None:
            model.load_state_dict(state_dict)
        return model

    def initialize(self, env_spaces, share_memory=False, **kwargs):
        """Build the model for ``env_spaces`` on the host (samplers call this before any worker is
        forked; ``share_memory`` keeps a fork-shared copy for CPU samplers, base.py:64-93)."""
        self.env_spaces, self.share_memory = env_spaces, share_memory
        self.env_model_kwargs = self.make_env_to_model_kwargs(env_spaces)
        self.model = self._new_model(self.initial_model_state_dict)
        if share_memory:
            self.shared_model = self.model.share_memory()
        if hasattr(env_spaces.action, "n"):       # discrete actions: the agent's action distribution
            self.distribution = self.make_distribution(env_spaces.action.n)

    def make_distribution(self, n_actions):
        """The distribution object of a discrete-action agent (None: the agent has none)."""
        return None

    def make_env_to_model_kwargs(self, env_spaces):
        return {}

    def to_device(self, cuda_idx=None):
        """Move the model to ``cuda:<cuda_idx>`` (None: stay on the host).  A fork-shared model
        stays behind for the CPU workers; the device gets its own copy."""
        if cuda_idx is None:
            return
        self.device = torch.device("cuda", index=cuda_idx)
        if self.shared_model is not None:
            self.model = self._new_model(self.shared_model.state_dict())
        self.model.to(self.device)
        logger.log(f"Initialized agent model on device: {self.device}.")

    def data_parallel(self, bucket_cap_mb=None):
        """Wrap the model in DistributedDataParallel: gradient all-reduce rides RCCL over
        xGMI when the process group backend is "nccl" (gloo on CPU)
        (rlpyt/agents/base.py:118-136).

        Bucketing: the models of this path are a few small tensors around ONE large one (the trunk
        weight: 7.1 MB of AtariFfModel's 7.14 MB) whose gradient is ready half a millisecond before
        the conv gradients.  With DDP's default 25 MB cap everything but the head shares one bucket,
        so the only sizeable all-reduce of a minibatch starts when backward has finished and is
        fully exposed.  A cap just below the largest tensor gives that tensor a bucket of its own:
        its all-reduce (xGMI ring, latency-class at this size) then runs under conv2_bwd /
        conv1_wgrad, and only the small tail bucket is exposed.  Same mean gradient either way."""
        from torch.nn.parallel import DistributedDataParallel as DDP
        device_id = self.device.index
        if bucket_cap_mb is None:
            sizes = [p.numel() * p.element_size() for p in self.model.parameters() if p.requires_grad]
            largest = max(sizes) / 2 ** 20 if sizes else 0.
            bucket_cap_mb = 25 if largest < 2 else max(1, min(25, int(0.6 * largest)))
        self.model = DDP(self.model, device_ids=None if device_id is None else [device_id],
                         output_device=device_id, bucket_cap_mb=bucket_cap_mb)
        logger.log(f"Initialized DistributedDataParallel agent model on device {self.device} "
                   f"(bucket cap {bucket_cap_mb} MB).")
        return device_id

    @property
    def sampling_model(self):
        """The bare module for no-grad sampling forwards: skips the DistributedDataParallel
        wrapper's per-call bookkeeping (and keeps the step hipGraph-capturable)."""
        return getattr(self.model, "module", self.model)

    @property
    def uses_prev_inputs(self):
        """False when the model ignores prev_action / prev_reward (e.g. AtariFfModel): the
        sampler then skips building them every step."""
        return getattr(self.sampling_model, "uses_prev_inputs", True)

    @property
    def supports_sample_uniforms(self):
        return False

    def collector_initialize(self, global_B=1, env_ranks=None):
        pass

    def select_envs(self, lo=None, hi=None):
        """Called by the sampler before it steps environments [lo, hi) of this rank (one pipeline
        group); None = all.  Agents with per-environment sampling state use it (vector epsilon)."""

    def step(self, observation, prev_action, prev_reward):
        raise NotImplementedError

    def step_into(self, observation, prev_action, prev_reward, binding):
        """Optional fast path of ``step`` for HBM-resident samplers: run the sampling forward
        AND write action[t+1] / agent_info[t] rows of the bound batch (``binding``: the sampler's
        ``StepBinding``) plus the host-bound action copy.  Return False to decline (the caller
        then uses ``step`` and commits the rows itself).  When ``binding.push`` is set (the
        sampler's ``FramePush``) the sampler has NOT yet rebuilt row ``t`` of the frame-stacked
        observation batch: an agent that accepts must do that too (``observation`` is then
        None); declining makes the sampler push the frames itself and ask again without it."""
        return False

    def reset(self):
        pass

    def reset_one(self, idx):
        pass

    def parameters(self):
        return self.model.parameters()

    def state_dict(self):
        return self.model.state_dict()

    def load_state_dict(self, state_dict):
        self.model.load_state_dict(state_dict)
        self._mode_itr = None                  # (derived weight buffers are stale: see ``_enter``)

    # The three modes differ in: the module's train flag, whether fused-step weight buffers need a
    # refresh (any mode the sampler steps the agent in), and what subclasses hook on (epsilon,
    # recurrent state): one table, one transition function.
    _MODES = {"train": (True, False), "sample": (False, True), "eval": (False, True)}

    def _enter(self, mode, itr):
        training, steps_envs = self._MODES[mode]
        # The runner AND the sampler both announce a sampling phase (``agent.sample_mode(itr)`` in
        # rlpyt/runners/minibatch_rl.py:237 and rlpyt/samplers/parallel/gpu/sampler.py:33): the second
        # call of a pair finds the agent already in that mode for that iteration and has nothing to
        # redo -- on config #3 the repeat cost a weight-packing launch, two epsilon uploads and a walk
        # over the module tree per iteration.  Anything that writes parameters in between (a state-dict
        # load, the asynchronous mailbox) clears ``_mode_itr``.
        repeat = LEAN_HOST and mode == self._mode and itr is not None and itr == self._mode_itr
        self._mode_repeat = repeat
        self._set_training(training)
        self._mode, self._mode_itr = mode, itr
        if steps_envs and not repeat:
            self._refresh_step_weights()

    def _set_training(self, training):
        """``model.train(training)`` without the per-call walk over the module tree (``Module.train``
        goes through ``named_children`` / ``__setattr__`` of every submodule: ~50 us for the DQN
        model, three times per iteration): the flag is written into the cached module list, and only
        when it differs."""
        model = self.model
        if not LEAN_HOST:
            model.train(training)
            return
        if model.training == training and getattr(self, "_train_flag_of", None) is model:
            return
        mods = getattr(self, "_train_mods", None)
        if mods is None or self._train_flag_of is not model:
            mods = self._train_mods = list(model.modules())
            self._train_flag_of = model
            model.train(training)              # the first time through the module's own method
            return
        for m in mods:
            m.training = training

    def train_mode(self, itr):
        self._enter("train", itr)

    def sample_mode(self, itr):
        self._enter("sample", itr)

    def eval_mode(self, itr):
        self._enter("eval", itr)

    def _refresh_step_weights(self):
        """Models that keep derived weight buffers for their fused sampling step (read by address
        from captured step graphs) bring them up to date with the parameters once per iteration."""
        m = getattr(getattr(self.model, "module", self.model), "refresh_step_weights", None)
        if m is not None:
            m()

    def sync_shared_memory(self):
        """Publish the trained parameters to the fork-shared copy the CPU workers read."""
        shared = self.shared_model
        if shared is None or shared is self.model:
            return
        shared.load_state_dict(strip_ddp_state_dict(self.model.state_dict()))

    def toggle_alt(self):
        pass

    # ---- asynchronous sampling / optimisation (rlpyt/agents/base.py:138-200: async_cpu,
    # send_shared_memory, recv_shared_memory) -------------------------------------------------------
    # The reference keeps a second copy of the model in OS shared memory: the optimizer process
    # publishes into it under a lock, the sampler process copies out of it between batches.  Sampler
    # side and optimizer side are threads of one process here and all three copies live in HBM: the
    # optimizer's model, the SAMPLER TWIN's model (the agent object the sampler steps: this agent's
    # settings, its own parameters, frozen during a batch) and the mailbox between them.  The copies
    # are device-to-device launches on the calling thread's stream, ordered across threads / streams
    # by a lock and two events.
    def async_twin(self):
        """The sampler-side agent of an asynchronous run (call after ``to_device``)."""
        import copy
        import threading
        twin = copy.copy(self)
        twin.model = copy.deepcopy(self.sampling_model)
        twin.shared_model = None
        twin._mode = twin._mode_itr = None
        twin._async_twin_reset()
        params = [p.detach() for p in self.sampling_model.parameters()]
        bufs = [b.detach() for b in self.sampling_model.buffers()]
        self._mailbox = twin._mailbox = dict(
            lock=threading.Lock(), version=0, written=None, read=None,
            tensors=[t.clone() for t in params + bufs])
        twin._mail_seen = 0
        return twin

    def _async_twin_reset(self):
        """Per-instance sampling state a twin must not share with its original."""

    def _mail_tensors(self):
        m = self.sampling_model
        return [p.detach() for p in m.parameters()] + [b.detach() for b in m.buffers()]

    @torch.no_grad()
    def send_shared_memory(self):
        """Optimizer side: publish the current parameters (after an ``optimize_agent`` call)."""
        box = getattr(self, "_mailbox", None)
        if box is None:
            return self.sync_shared_memory()
        with box["lock"]:
            cuda = self.device.type == "cuda"
            if cuda and box["read"] is not None:          # the sampler's last copy-out is done first
                torch.cuda.current_stream(self.device).wait_event(box["read"])
            torch._foreach_copy_(box["tensors"], self._mail_tensors())
            box["version"] += 1
            if cuda:
                box["written"] = torch.cuda.Event()
                box["written"].record(torch.cuda.current_stream(self.device))

    @torch.no_grad()
    def recv_shared_memory(self):
        """Sampler side: take over newly published parameters (between batches); True if there were."""
        box = getattr(self, "_mailbox", None)
        if box is None:
            return False
        with box["lock"]:
            if box["version"] == getattr(self, "_mail_seen", 0):
                return False
            cuda = self.device.type == "cuda"
            if cuda and box["written"] is not None:
                torch.cuda.current_stream(self.device).wait_event(box["written"])
            mine = self._mail_tensors()
            torch._foreach_copy_(mine, box["tensors"])      # (bumps the parameters' version counters)
            self._mode_itr = None                           # (see ``_enter``: derived buffers are stale)
            self._mail_seen = box["version"]
            if cuda:
                box["read"] = torch.cuda.Event()
                box["read"].record(torch.cuda.current_stream(self.device))
        return True

    def gather_observation(self, observation, flat_idx):
        """Minibatch rows ``idx -> (idx % T, idx // T)`` of a [T,B,...] observation batch
        (rlpyt/algos/pg/ppo.py:94-100); image agents override this to deliver the rows
        already converted for their conv stack."""
        from .. import ops
        return ops.gather_tb(observation.contiguous(), flat_idx)

    # -- helpers ------------------------------------------------------------------------
    def _to_model_device(self, *xs):
        return tuple(x if (x is None or not isinstance(x, torch.Tensor) or x.device == self.device)
                     else x.to(self.device, non_blocking=True) for x in xs)

    def _out(self, x):
        """Leave outputs in HBM unless running in reference drop-in mode."""
        from ..utils.buffer import buffer_to
        return buffer_to(x, device="cpu") if self.host_outputs else x


class RecurrentAgentMixin:
    """Recurrent state management during sampling (rlpyt/agents/base.py:252-304), re-shaped for
    the HBM-resident sampler: the state of every pipeline group (``select_slot``) is ONE
    persistent ``[N, B_g, H]`` device buffer updated in place, so the step can live in a captured
    hipGraph; resets after ``done`` are a masked multiply on the device (``reset_where``) instead
    of per-environment host indexing (``reset_one``, kept for API compatibility)."""
    recurrent = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._rnn_states = {}
        self._stash = None
        self._slot = 0

    def select_slot(self, slot):
        self._slot = slot

    @property
    def prev_rnn_state(self):
        return self._rnn_states.get(self._slot)

    def advance_rnn_state(self, new_rnn_state):
        from ..utils.buffer import buffer_func, buffer_leaves
        cur = self._rnn_states.get(self._slot)
        if cur is None:
            self._rnn_states[self._slot] = buffer_func(new_rnn_state, lambda x: x.clone())
        else:
            for d, s in zip(buffer_leaves(cur), buffer_leaves(new_rnn_state)):
                d.copy_(s)

    def reset(self):
        self._rnn_states = {}

    def reset_one(self, idx):
        from ..utils.buffer import buffer_leaves
        cur = self._rnn_states.get(self._slot)
        if cur is not None:
            for x in buffer_leaves(cur):
                x[:, idx] = 0

    def _async_twin_reset(self):
        self._rnn_states, self._stash, self._slot = {}, None, 0

    def reset_where(self, mask):
        """Zero the state of the environments where ``mask`` ([B_g] bool, same device)."""
        from ..utils.buffer import buffer_leaves
        cur = self._rnn_states.get(self._slot)
        if cur is not None:
            for x in buffer_leaves(cur):
                x.mul_((~mask).reshape(1, -1, 1).to(x.dtype))

    def _enter(self, mode, itr):
        """Sampling state survives the excursions into training / evaluation: it is parked when the
        agent leaves sample mode and restored when it comes back; train and eval start stateless."""
        if mode == "sample":
            if self._mode != "sample" and self._stash is not None:
                self._rnn_states = self._stash
        else:
            if self._mode == "sample":
                self._stash = self._rnn_states
            self._rnn_states = {}
        super()._enter(mode, itr)
