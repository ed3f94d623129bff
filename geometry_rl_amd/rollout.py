"""The once-per-rollout part of the training loop around the policy update (examples/torchrl/train.py:249-316):

    values over the [N, T+1] frames  ->  shifted GAE  ->  ppo_epochs x (N*T / mini_batch) minibatch updates, each minibatch
    drawn WITHOUT replacement.

The reference keeps the rollout in a CPU ``LazyTensorStorage`` and moves every minibatch to the device (train.py:120,128,261);
here the rollout stays resident in HBM (4096 x 128 rigid frames are 1.6 GB) and a minibatch is assembled by ONE gather launch
(``grl_gather_rows_many``) straight into the static input buffers of the recorded update step.

Sampling (``RolloutDriver(mini_batch_size=, sampling=)``), always without replacement, every frame exactly once per epoch:

* ``"env_aligned"`` (default): per environment a random permutation of its T time steps; a minibatch of ``k * N`` frames
  (``mini_batch_size`` must be a multiple of the number of environments N; k = 1 for rigid / rope whose configs set
  mini_batch_size = num_envs, k = 2 for cloth: configs/cloth_hanging_multi_hepi_trpl_cfg.yaml:40,125) takes k permutation columns,
  laid out column after column, so row i of EVERY minibatch belongs to environment i mod N.  That is what the graph topology
  cached per batch size assumes (rigid_tasks_data.py:254-255: valid point counts and kNN edges are those of the first batch of
  that size).  DEVIATION from the reference, which draws uniformly from the flattened N*T frames (SamplerWithoutReplacement,
  train.py:128) and therefore pairs its cached placeholders with rows of other environments -- harmless only when all
  environments share one topology.
* ``"uniform"``: the reference's sampler -- a random permutation of the N*T frames cut into minibatches of ``mini_batch_size``
  (last one dropped if short: the recorded step has one size).  Allowed for task families whose topology does not depend on the
  row (cloth: fully connected hole boundary, fixed sizes); for rigid (ragged point counts) it raises unless
  ``allow_stale_topology=True`` reproduces the reference quirk knowingly."""
from typing import Dict, Iterator, Optional

import torch

from . import agent as _agent

PPO_KEYS = ("action", "loc", "var", "sample_log_prob", "state_value", "advantage", "value_target")


class RolloutBuffer:
    """Device-resident rollout: every tensor is [N, T, ...]; ``flat(k)`` views it as rows [N*T, width]."""

    def __init__(self, data: Dict[str, torch.Tensor]):
        some = next(iter(data.values()))
        self.N, self.T = some.shape[0], some.shape[1]
        self.data = {k: v.contiguous() for k, v in data.items()}
        for k, v in self.data.items():
            if v.shape[:2] != (self.N, self.T):
                raise ValueError(f"{k}: expected leading dims [{self.N}, {self.T}], got {tuple(v.shape)}")

    def flat(self, k: str) -> torch.Tensor:
        v = self.data[k]
        return v.reshape(self.N * self.T, -1)

    def rows(self, idx: torch.Tensor, keys) -> Dict[str, torch.Tensor]:
        """Plain (allocating) minibatch: rows ``idx`` of the given keys."""
        return {k: self.flat(k).index_select(0, idx) for k in keys}


class RolloutDriver:
    def __init__(self, updater: "_agent.PolicyUpdater", spec, gamma: float = 0.99, lmbda: float = 0.95, ppo_epochs: int = 5,
                 seed: int = 0, mini_batch_size: Optional[int] = None, sampling: str = "env_aligned",
                 allow_stale_topology: bool = False):
        self.updater, self.spec = updater, spec
        self.gamma, self.lmbda, self.ppo_epochs = gamma, lmbda, ppo_epochs
        self.gen = None
        self.seed = seed
        self.mini_batch_size = mini_batch_size   # None: one frame per environment (= num_envs)
        if sampling not in ("env_aligned", "uniform"):
            raise ValueError(sampling)
        # raggedness, not the task family, is what makes the cached topology wrong for rows of other environments: rigid tasks carry a
        # per-sample object_num_points, variable-length ropes a per-sample links_num_points (padded nodes are dropped by the CACHED counts)
        infos = (getattr(spec, "obs_names", None) or {}).get("infos", ())
        ragged = any(n in infos for n in ("object_num_points", "links_num_points"))
        if sampling == "uniform" and ragged and not allow_stale_topology:
            raise ValueError("uniform sampling pairs the topology cached per batch size with rows of other environments; this task has "
                             "ragged per-sample point counts (object_num_points / links_num_points): pass allow_stale_topology=True to "
                             "reproduce the reference quirk")
        self.sampling = sampling

    # ---- train.py:134-140,249-251: critic over the T+1 frames of every environment, then the shifted GAE scan
    @torch.no_grad()
    def compute_advantages(self, buf: RolloutBuffer, next_last: Dict[str, torch.Tensor]) -> None:
        """``next_last``: observation groups of the frame after the last one, [N, 1, width].  Writes ``state_value``,
        ``advantage`` and ``value_target`` [N, T, 1] into the buffer."""
        critic = self.updater.loss_module.critic_network
        N, T = buf.N, buf.T
        vals = torch.empty(N, T + 1, device=buf.data["reward"].device, dtype=torch.float32)
        # all T frames of every environment as ONE time-batched critic call (policy.GNNVFNet._values_time_batched: the time steps are groups
        # of one launch set, statistics per step; data parallel: the per-step statistics of a chunk travel in one all-reduce per LayerNorm
        # stage -- two collectives per chunk, where the looped form issued two per time step), the frame after the last one as a second call
        vals[:, :T] = critic(*[buf.data[k] for k in self.spec.in_features], train=False).reshape(N, T)
        vals[:, T:] = critic(*[next_last[k] for k in self.spec.in_features], train=False).reshape(N, 1)
        adv, tgt = _agent.gae(buf.data["reward"].reshape(N, T), buf.data["done"].reshape(N, T), buf.data["terminated"].reshape(N, T),
                              vals, self.gamma, self.lmbda)
        buf.data["state_value"] = vals[:, :T].reshape(N, T, 1).contiguous()
        buf.data["advantage"] = adv.reshape(N, T, 1)
        buf.data["value_target"] = tgt.reshape(N, T, 1)

    # ---- sampler without replacement (train.py:128,258)
    def epoch_indices(self, N: int, T: int, device) -> torch.Tensor:
        """[T, N] int64 row indices into the flattened [N*T] rollout: row j = minibatch j."""
        if self.gen is None:
            self.gen = torch.Generator(device=device)
            self.gen.manual_seed(self.seed)
        perm = torch.argsort(torch.rand(N, T, device=device, generator=self.gen), dim=1)       # per-environment permutation of time
        return (torch.arange(N, device=device)[:, None] * T + perm).t().contiguous()

    def epoch_minibatches(self, N: int, T: int, device):
        """List of int64 index tensors into the flattened [N*T] rollout, one per minibatch of this epoch."""
        mbs = self.mini_batch_size or N
        if self.sampling == "uniform":
            if self.gen is None:
                self.gen = torch.Generator(device=device)
                self.gen.manual_seed(self.seed)
            perm = torch.randperm(N * T, device=device, generator=self.gen)
            return [perm[i:i + mbs] for i in range(0, N * T - mbs + 1, mbs)]
        if mbs % N:
            raise ValueError(f"env-aligned sampling needs mini_batch_size ({mbs}) to be a multiple of the number of environments ({N})")
        k = mbs // N
        idx = self.epoch_indices(N, T, device)                    # [T, N]
        return [idx[j:j + k].reshape(-1) for j in range(0, T - k + 1, k)]   # k columns, column after column

    def publish_advantage_stats(self, buf: RolloutBuffer, idxs) -> None:
        """Data parallel: the batch normalisation of the advantages (trpl.py:248-252) needs the (sum, sum of squares) over ALL ranks'
        shares of a minibatch.  The minibatches of an epoch are known when it starts, so their statistics are reduced by ONE all-reduce
        per epoch ([n_minibatches, 2] fp64) instead of one per update, and written next to the advantages as a per-frame column
        ``adv_stats`` (every frame carries its minibatch's global sums): the update's row gather brings them along and the fused loss
        kernel reads row 0 -- no statistics kernel, no collective and no graph boundary for them inside the update
        (``PolicyUpdater._plan``).  Rebuilt at every epoch (the sampler reshuffles)."""
        upd = self.updater
        if upd.group is None or not upd.loss_module.normalize_advantage:
            return
        idx = torch.stack([i.reshape(-1) for i in idxs])                           # [n_mb, frames per rank]
        a = buf.flat("advantage").reshape(-1).double()[idx]
        s = torch.stack([a.sum(1), (a * a).sum(1)], dim=1).contiguous()
        upd._reduce("sum", s, "advantage_stats_epoch")
        if "adv_stats" not in buf.data:
            buf.data["adv_stats"] = torch.zeros(buf.N, buf.T, 2, device=s.device, dtype=torch.float64)
        buf.flat("adv_stats")[idx.reshape(-1)] = s.repeat_interleave(idx.shape[1], dim=0)

    def minibatches(self, buf: RolloutBuffer) -> Iterator[torch.Tensor]:
        dev = next(iter(buf.data.values())).device
        for _ in range(self.ppo_epochs):
            idxs = self.epoch_minibatches(buf.N, buf.T, dev)
            self.publish_advantage_stats(buf, idxs)
            for idx in idxs:
                yield idx

    def run(self, buf: RolloutBuffer, next_last: Optional[Dict[str, torch.Tensor]] = None):
        """One rollout pass: [GAE] + ppo_epochs * T policy updates.  Returns the loss dict of the last update."""
        if next_last is not None:
            self.compute_advantages(buf, next_last)
        out = None
        dev = next(iter(buf.data.values())).device
        for _ in range(self.ppo_epochs):
            idxs = self.epoch_minibatches(buf.N, buf.T, dev)
            self.publish_advantage_stats(buf, idxs)
            if idxs and all(i.numel() == idxs[0].numel() for i in idxs) and hasattr(self.updater, "run_minibatches"):
                # the epoch's minibatches in one call: one rank with recorded lanes takes several steps per launch (PolicyUpdater.run_minibatches)
                out = self.updater.run_minibatches(buf, torch.stack([i.reshape(-1) for i in idxs]))
            else:
                for idx in idxs:
                    out = self.updater.step_from(buf, idx)
        return out


class PolicyActor:
    """Collector-side actor: ``ProbabilisticActor(TensorDictModule(policy, in_keys, ["loc", "covariance_matrix"]),
    distribution_class=MultivariateNormal, return_log_prob=True, default_interaction_type=RANDOM)`` of
    examples/torchrl/builders/utils_algo_graph.py:146-158 -- one no-grad policy pass per environment step (train.py:114-123) with the
    same forward kernels as the update, followed by the sampling kernel.  ``__call__(obs)`` returns the keys the collector writes
    into the rollout: ``loc``, ``var`` (the diagonal of covariance_matrix), ``action``, ``sample_log_prob``.

    ``use_graph``: after the first call (which builds the cached topology and, on a fresh policy, calibrates the convolutions --
    the collector calls the module with its default ``train=True``, gnn_gaussian_policy_diag.py:65) the pass is recorded into a
    hipGraph and replayed on static input buffers."""

    def __init__(self, policy, spec, use_graph: bool = True, seed: int = 0, deterministic: bool = False):
        self.policy, self.spec, self.use_graph, self.deterministic = policy, spec, use_graph, deterministic
        self.gen, self.seed = None, seed
        self._graph, self._static, self._out, self._calls = None, None, None, 0

    def _pass(self, obs):
        from . import hip
        loc, sigma = self.policy.forward_diag(*[obs[k] for k in self.spec.in_features], train=True)
        B, A = loc.shape
        eps = torch.zeros_like(loc) if self.deterministic else torch.randn(loc.shape, device=loc.device, dtype=loc.dtype, generator=self.gen)
        action, var = torch.empty_like(loc), torch.empty_like(loc)
        logp = torch.empty(B, device=loc.device, dtype=torch.float32)
        hip.call("grl_gaussian_sample", loc.contiguous(), sigma.contiguous(), eps, action, logp, var, B, A)
        return {"loc": loc, "var": var, "action": action, "sample_log_prob": logp}

    @torch.no_grad()
    def __call__(self, obs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        dev = obs[self.spec.in_features[0]].device
        if self.gen is None and not self.deterministic:
            self.gen = torch.Generator(device=dev)
            self.gen.manual_seed(self.seed)
        self._calls += 1
        if not self.use_graph or self._calls == 1:
            return self._pass(obs)
        if self._graph is None:
            self._static = {k: obs[k].clone() for k in self.spec.in_features}
            g = torch.cuda.CUDAGraph()
            if self.gen is not None and hasattr(g, "register_generator_state"):
                g.register_generator_state(self.gen)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with _agent._no_gc_while_capturing(), torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                self._out = self._pass(self._static)
            torch.cuda.current_stream().wait_stream(side)
            self._graph = g
        for k in self.spec.in_features:
            self._static[k].copy_(obs[k])
        self._graph.replay()
        return self._out


def collect(env_step, first_obs: Dict[str, torch.Tensor], actor: PolicyActor, T: int, normalizer=None):
    """The data-collection half of one training iteration (train.py:114-123, 232-247) with everything on the device: for T steps the
    (optionally normalised) observation goes through the collector-side actor, the action into ``env_step(action) -> (next raw
    observation groups, reward [N], done [N] bool, terminated [N] bool)``; the frames are stacked into a ``RolloutBuffer`` [N, T, ...].
    Returns ``(buffer, next_last)`` as ``RolloutDriver.run`` takes them (``next_last``: the observation after the last step, [N, 1, width])."""
    raw = first_obs
    frames = []
    for _ in range(T):
        obs = normalizer(raw) if normalizer is not None else raw
        out = actor(obs)
        rec = {k: v.clone() for k, v in obs.items()}
        rec.update({k: out[k].clone() for k in ("loc", "var", "action", "sample_log_prob")})
        raw, reward, done, terminated = env_step(out["action"])
        rec.update(reward=reward.reshape(-1, 1).float(), done=done.reshape(-1, 1), terminated=terminated.reshape(-1, 1))
        frames.append(rec)
    last = normalizer(raw, update=False) if normalizer is not None else raw
    data = {k: torch.stack([f[k] for f in frames], dim=1) for k in frames[0]}
    return RolloutBuffer(data), {k: v.unsqueeze(1) for k, v in last.items()}
