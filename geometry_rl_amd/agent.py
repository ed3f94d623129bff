"""Wiring of actor, critic, projection and loss for one task config -- the counterpart of
``examples/torchrl/builders/agent.py:14-80`` + ``builders/utils_algo_graph.py:208-276`` without Hydra / the Isaac env object
(observation layout comes from a TaskSpec instead of ``env.observation_manager``), plus the policy-update driver that
replaces the inner loop of ``examples/torchrl/train.py:258-316``."""
from dataclasses import dataclass
from typing import Dict, Optional

import contextlib
import os

import torch
import torch.nn as nn

from . import hip, ops
from .graph import HyperData, TaskSpec
from .hepi import HEPi, FiberBundleConv
from .policy import BaseCritic, DeepSets, GNNGaussianPolicyDiag, GNNVFNet
from .trpl import KLProjectionLayer, TRPLLoss


@dataclass
class AgentConfig:
    """Values of configs/<task>_hepi_trpl_cfg.yaml that reach the hot path."""
    model: str = "hepi"
    dim: int = 3
    num_ori: int = 16
    only_upper_hemisphere: bool = False
    output_dim: int = 1
    output_dim_vec: int = 1
    num_layers: int = 2
    codes: tuple = ((1, 0), (0, 1), (0, 1))  # configs/algorithm/pyg_agent/model/hepi.yaml:17-48
    init_std: float = 1.0
    minimal_std: float = 1e-5
    mean_bound: float = 0.05
    cov_bound: float = 0.0025
    proj_type: str = "kl"  # kl | frob | w2
    trust_region_coeff: float = 1.0
    entropy_coef: float = 0.005
    critic_coef: float = 0.5
    clip_value: float = 0.2
    lr: float = 3e-4
    clip_grad_norm: bool = False
    max_grad_norm: float = 1.0
    aggr: str = "add"         # "AttentionalAggregation": configs/algorithm/pyg_agent/model/hepi_attention.yaml
    precision: str = "fp32"   # "bf16": BASELINE config 5 -- node latents stored as bf16, one bf16 MFMA per dense product, fp32 accumulation


def build_agent(spec: TaskSpec, cfg: AgentConfig, device="cuda", group=None):
    """-> (actor GNNGaussianPolicyDiag, critic BaseCritic, projection, loss_module)  (agent.py:31-52)."""
    n_in = len(spec.node_types) + spec.n_vec  # utils_algo_graph.py:79
    if cfg.model == "hepi":
        mp = []  # utils_algo_graph.py:29-47: one fresh conv per (level, active round)
        for lvl in range(len(spec.edge_levels)):
            mp.append([FiberBundleConv(64, 64, 64, groups=64, separable=True, widening_factor=4, aggr=cfg.aggr) if cfg.codes[lvl][k] else None
                       for k in range(len(cfg.codes[lvl]))])
        gnn = HEPi(input_dim_node=n_in, input_dim_edge=len(spec.edge_types) + 4, hidden_dim=64, latent_dim=64,
                   output_dim=cfg.output_dim, output_dim_vec=cfg.output_dim_vec, node_type_mapping=spec.node_types,
                   edge_type_mapping=[tuple(e) for e in spec.edge_types], edge_level_mapping=spec.edge_levels,
                   message_passing=mp, num_messages=len(cfg.codes[0]), device=device, num_ori=cfg.num_ori,
                   ponita_dim=cfg.dim, only_upper_hemisphere=cfg.only_upper_hemisphere, precision=cfg.precision)
    elif cfg.model == "transformer":   # BASELINE config 1: stock-torch baseline actor + post_fc head on the same loss / critic / updater
        from .transformer import TransformerVanilla
        gnn = TransformerVanilla(input_dim_node=len(spec.node_types) + 3 * spec.n_vec, output_dim=64, num_layers=cfg.num_layers,
                                 num_heads=2, hidden_dim=64, dropout=0.0, concat_global=False, device=device)
    elif cfg.model == "empn":
        from .ponita_gcn import PonitaGCN
        gnn = PonitaGCN(input_dim_node=n_in, output_dim=cfg.output_dim, output_dim_vec=cfg.output_dim_vec,
                        num_layers=cfg.num_layers, hidden_dim=64, num_ori=cfg.num_ori, ponita_dim=cfg.dim,
                        only_upper_hemisphere=cfg.only_upper_hemisphere, device=device, precision=cfg.precision)
    else:
        raise ValueError(cfg.model)
    post_fc = cfg.model == "transformer"
    a_data = HyperData(spec, full_graph_obs=False, dist_as_pos=True, output_mask_key=spec.actuator, concat_input_vector=post_fc)
    A = spec.num_actuators * cfg.output_dim_vec * 3
    actor = GNNGaussianPolicyDiag(gnn=gnn, hyper_data=a_data, action_dim=A, num_actuators=spec.num_actuators, init="orthogonal",
                                  hidden_sizes=(64, 64), contextual_std=True, init_std=cfg.init_std, minimal_std=cfg.minimal_std,
                                  share_action_dim=True, post_fc=post_fc)
    actor.group = group
    c_data = HyperData(spec, full_graph_obs=True, dist_as_pos=False, output_mask_key=None, concat_input_vector=True)
    c_gnn = DeepSets(input_dim_node=len(spec.node_types) + 3 * spec.n_vec, output_dim=64, hidden_dim=64, device=device)
    critic = BaseCritic(GNNVFNet(gnn=c_gnn, hyper_data=c_data))
    critic._network1.group = group
    projection = KLProjectionLayer(proj_type=cfg.proj_type, mean_bound=cfg.mean_bound, cov_bound=cfg.cov_bound,
                                   trust_region_coeff=cfg.trust_region_coeff, scale_prec=True, entropy_schedule=False, action_dim=A)
    # config 1 hands its actor the NORMALISED vectors in the raw-vector slots (configs/rigid_insertion_multi_transformer_trpl_cfg.yaml:88-94)
    a_in = [k if k.startswith("norm_") or "vectors" not in k else "norm_" + k for k in spec.in_features] if post_fc else spec.in_features
    loss = TRPLLoss(actor, critic, projection=projection, entropy_coef=cfg.entropy_coef, critic_coef=cfg.critic_coef,
                    clip_value=cfg.clip_value, loss_critic_type="l2", normalize_advantage=True, in_features=a_in,
                    critic_in_features=spec.in_features, group=group)
    return actor, critic, projection, loss


_CRITIC_GROUPS = {}   # (id of the actor lane's process group, its ranks) -> the critic lane's communicator


@contextlib.contextmanager
def _no_gc_while_capturing():
    """Python's cyclic collector must not run while a stream is capturing: if it frees an object that owns device resources -- the hipGraphs
    or events of an updater that went out of use -- their destruction inside the capture is an error raised from a destructor, and the
    process aborts (seen once in six runs of tests/test_gpu_rollout.py: "Fatal Python error: Aborted ... Garbage-collecting" under
    _compile_epoch; this torch's ``torch.cuda.graph`` no longer collects on entry).  Garbage is collected BEFORE the capture, and the
    collector is held for its duration."""
    import gc
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


class PolicyUpdater:
    """One policy-update step = loss forward, actor + critic backward, optional clip_grad_norm_ per network, two Adam(lr,
    eps=1e-5) steps (train.py:279-316).  Parameters of both networks live in ONE flat fp32 buffer (gradients likewise), so a
    data-parallel run needs a single RCCL all-reduce of the gradient per step and Adam is a single kernel per optimizer.

    The step is laid out as an explicit program (``_plan``) of device-only segments, each free of host synchronisation, so with
    ``use_graph=True`` they are recorded once into hipGraphs (torch.cuda.CUDAGraph) and replayed -- the ~3 ms of per-step launch
    overhead disappears, which is what strong scaling over 8 GPUs needs (512 frames per GPU are < 1 ms of device time).  One program per
    case (``_plan_one_stream`` / ``_plan_lanes`` / ``_plan_dp``): see the comment above ``_plan``."""

    def __init__(self, loss_module: TRPLLoss, lr=3e-4, eps=1e-5, betas=(0.9, 0.999), clip_grad_norm=False, max_grad_norm=1.0,
                 group=None, use_graph=False, overlap_critic=True, allow_eager_fallback=False, force_dp_plan=False,
                 critic_after_first_conv=True):
        self.loss_module, self.group = loss_module, group
        self.overlap_critic = overlap_critic   # one rank: False = everything on the caller's stream (_plan_one_stream)
        self.force_dp_plan = force_dp_plan     # a process group of ONE rank runs the data-parallel program (bench.py --dp-plan)
        self.critic_delay_us = 0               # experiment knob (tools/critic_delay_ab.py): microseconds the critic's lane idles before it starts
        # one rank, two lanes: the critic's lane starts when the actor's FIRST edge convolution has finished (a stream wait on a flag that a
        # 4-byte copy behind that launch sets to the step count: _plan_lanes).  Beside that launch the critic's kernels cost it 60-90 us at 4096
        # frames (DESIGN.md finding 42); beside the HBM-bound kernels that follow they cost less: -1.6 % on the step (finding 55).
        self.critic_after_first_conv = critic_after_first_conv
        # the critic lane's gate: a launch of the lane itself (grl_wait_flag_ge, default) or hipStreamWaitValue32 in front of its graph
        # (round 5; GRL_GATE_STREAMWAIT=1 -- a BETA API: taken only where the device reports support)
        self.gate_in_graph = os.environ.get("GRL_GATE_STREAMWAIT", "0") != "1"
        self.epoch_unroll = int(os.environ.get("GRL_EPOCH_UNROLL", "8"))   # minibatch steps per recorded launch of run_minibatches
        # data parallel: gate the critic's lane behind the actor's first edge convolution from this shard size on (0: never).  One-rank RCCL
        # group, alternating on one box (profiles/r06_ab_dp_gate.txt): -4.1 % at 4096 frames per rank, -2 % at 2048, -1.3 % at 1024, 0 at 512
        self.dp_gate_from_frames = int(os.environ.get("GRL_DP_GATE_FROM", "1024"))
        self.dp_eager_tail = os.environ.get("GRL_DP_EAGER_TAIL", "1") == "1"   # the actor lane's one-launch tail behind the collective: a plain launch, not a one-node graph
        # gated sizes: one step per launch with the gathers inside (by device cursor).  Measured no better than the per-step program with its
        # eager gather (256 / 512 frames: -0.5 % / +0.5 %) and 1 % slower at 4096 frames (gpurun_out -> profiles/r06_ab_forms.txt): OFF
        self.epoch_cursor = os.environ.get("GRL_EPOCH_CURSOR", "0") == "1"
        self.epoch_unroll_max_gated_frames = int(os.environ.get("GRL_EPOCH_UNROLL_MAX_GATED", "64"))   # ... above this many frames
        self.epoch_gated_from_frames = int(os.environ.get("GRL_EPOCH_GATED_FROM", "3072"))   # run_minibatches: the gated per-step program from here on
        self._epoch = None
        # which recorded form run_minibatches takes above 64 work-frames is measured once per size (_tune_form); GRL_AUTOTUNE_FORM=0: the table
        self.autotune_form = os.environ.get("GRL_AUTOTUNE_FORM", "1") == "1"
        self.form_by_size, self.form_times = {}, {}

        self.allow_eager_fallback = allow_eager_fallback   # False: a failed hipGraph capture raises instead of degrading silently
        self.mode = "graph" if use_graph else "eager"      # what actually runs (bench.py reports it)
        self._hyper = dict(eps=eps, betas=tuple(betas), clip=clip_grad_norm, max_norm=max_grad_norm)
        a = [p for p in loss_module.actor_network.parameters() if p.requires_grad]
        c = [p for p in loss_module.critic_network.parameters() if p.requires_grad]
        self.params = a + c
        pad4 = lambda k: (k + 3) & ~3   # every parameter starts 16-byte aligned (vector loads in the weight-staging prologues)
        self.n_actor = sum(pad4(p.numel()) for p in a)
        n = sum(pad4(p.numel()) for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        if not self.gate_in_graph and self.flat.is_cuda and not hip.query("grl_can_stream_wait_value"):
            self.gate_in_graph = True   # (no hipStreamWaitValue32 on this device: the gate as a launch)
        # the flat gradient, with room IN FRONT of it for the ranks' loss records ([world][14] (hi, lo) float pairs, grl_trpl_fold_record_pairs):
        # data parallel, ``gbuf[:rec + n_actor]`` is ONE all-reduce -- the records ride on the actor's gradient slice
        if group is not None:
            import torch.distributed as dist
            self.rank, n_ranks = dist.get_rank(group), dist.get_world_size(group)
        else:
            self.rank, n_ranks = 0, 1
        self._rec = 28 * n_ranks
        self.gbuf = torch.zeros(self._rec + n, device=dev, dtype=torch.float32)
        self.gflat = self.gbuf[self._rec:]
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.gflat[off:off + k].view_as(p)
            off += pad4(k)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)  # optimizer step, device side (graph replays)
        self.step_dev_c = torch.zeros(1, device=dev, dtype=torch.int32)  # the same count kept by the critic's lane
        self.lane_flag = torch.zeros(1, device=dev, dtype=torch.int32)   # step count of the last "first edge convolution finished" signal
        # learning rate, device side: the recorded Adam launches read it, so an annealed rate (train.py:264-271 writes
        # ``group["lr"] = lr * alpha`` before every iteration; configs/algorithm/optim/default.yaml:5) takes effect under replay
        self.lr_dev = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
        self._lr = float(lr)
        self.use_graph = use_graph
        if use_graph and getattr(loss_module.actor_network, "post_fc", False):
            # config 1's baseline actor is a stock torch.nn.TransformerEncoder: its launches are torch's own (rocBLAS / hipBLASLt
            # workspaces are not capture-safe on this stack), so that step is issued eagerly -- it is the reference's CPU-sized
            # plumbing case (64 envs x 32 steps), not a throughput path
            self.use_graph, self.mode = False, "eager (stock-torch transformer actor: not recorded)"
        self._static = None
        self._program = None
        # Leaf gradients are WRITTEN by the one fold launch at the end of the backward pass (ops.flush_deferred_grads(overwrite=True)) instead
        # of accumulated into a zeroed buffer: no per-step zeroing launch.  Valid while EVERY leaf gradient of the step comes through the
        # fold queue: not with the stock torch transformer actor or the attention gate (torch's AccumulateGrad adds into .grad) -- and
        # checked on the first eager step of every updater (``_check_overwrite_coverage``).
        gnn = getattr(loss_module.actor_network, "gnn", None)
        self._fold_overwrite = (not getattr(loss_module.actor_network, "post_fc", False)
                                and not any(getattr(mod, "attention", False) for mod in (gnn.modules() if gnn is not None else [])))
        self._overwrite_checked = not self._fold_overwrite
        # the critic's lane reduces on a communicator of its own (_plan_dp).  dist.new_group is collective over the WHOLE default group
        # and must be entered by every rank in the same order: ``group`` therefore has to span WORLD (asserted), and a failure propagates
        # -- ranks that disagree on which communicator carries the critic's collectives would deadlock in the first step.
        # GRL_DP_ONE_COMM=1 (the documented fallback, README "switches"): both lanes on ``group``.
        self.group_c = None
        if group is not None and os.environ.get("GRL_DP_ONE_COMM", "0") == "0":
            import torch.distributed as dist
            ranks = dist.get_process_group_ranks(group)
            if len(ranks) != dist.get_world_size():
                raise ValueError("PolicyUpdater(group=...) must span the default process group (dist.new_group for the critic's lane is "
                                 "collective over WORLD); set GRL_DP_ONE_COMM=1 to run both lanes on a sub-group's own communicator")
            # ONE extra communicator per process group, shared by every updater built on it (a communicator per updater was
            # never destroyed -- the data-parallel tests leaked one each)
            key = (id(group), tuple(ranks))
            if key not in _CRITIC_GROUPS:
                _CRITIC_GROUPS[key] = dist.new_group(ranks=ranks)
            self.group_c = _CRITIC_GROUPS[key]
        # GRL_DP_ONESHOT=1: the actor lane's one collective (gradient slice + loss records) as a one-shot all-reduce over hipIpc-mapped peer
        # buffers (geometry_rl_amd/oneshot.py) instead of RCCL.  OFF by default: tested with stand-in ranks on one GPU only
        # (tests/test_gpu_oneshot.py); DESIGN.md section 5 has the switch-on criterion for a real node.
        self._oneshot = None
        if group is not None and n_ranks > 1 and os.environ.get("GRL_DP_ONESHOT", "0") == "1" and self.flat.is_cuda:
            import torch.distributed as dist
            if dist.get_backend(group) == "nccl":
                from . import oneshot
                self._oneshot = oneshot.ipc_rank(group, self.gbuf[:self._rec + self.n_actor])
        if group is not None:
            self.sync_replicas()

    # ---- hyper-parameters.  ``lr`` lives in device memory (no re-recording); the others are baked into recorded launches as
    #      scalars, so changing one drops the recorded program (it is re-recorded by the next step)
    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float):
        if float(value) != self._lr:
            self._lr = float(value)
            self.lr_dev.fill_(self._lr)

    def _set_hyper(self, key, value):
        if self._hyper[key] != value:
            self._hyper[key] = value
            self._program, self._epoch = None, None   # recorded launches carry the old scalar

    eps = property(lambda self: self._hyper["eps"], lambda self, v: self._set_hyper("eps", v))
    betas = property(lambda self: self._hyper["betas"], lambda self, v: self._set_hyper("betas", tuple(v)))
    clip = property(lambda self: self._hyper["clip"], lambda self, v: self._set_hyper("clip", bool(v)))
    max_norm = property(lambda self: self._hyper["max_norm"], lambda self, v: self._set_hyper("max_norm", float(v)))

    def anneal_lr(self, base_lr: float, iteration: int, total_iterations: int) -> float:
        """train.py:264-271: ``alpha = 1 - i / total; lr = base_lr * alpha`` for both optimisers."""
        self.lr = base_lr * (1.0 - iteration / float(total_iterations))
        return self.lr

    def sync_replicas(self):
        """Data parallel: every replica takes rank 0's parameters AND its ``callibrated`` latches (the data-dependent
        re-initialisation of conv.py:104-105 is rank-local arithmetic on rank-local data; replicas must not each run their own)."""
        if self.group is None:
            return
        import torch.distributed as dist
        src = dist.get_global_rank(self.group, 0) if hasattr(dist, "get_global_rank") else 0
        dist.broadcast(self.flat, src=src, group=self.group)
        actor = self.loss_module.actor_network
        flags = [b for n, b in actor.named_buffers() if n.endswith("callibrated")]
        if flags:
            t = torch.stack([f.to(torch.uint8) for f in flags]).to(self.flat.device)
            dist.broadcast(t, src=src, group=self.group)
            for f, v in zip(flags, t):
                f.fill_(bool(v))
        actor._calib_checked = False   # re-inspect the latches on the next training forward

    # ---- the step's programs.  A program is a list of ("run", fn [, lane]) | ("sum", tensor getter, lane, label) | ("fork" | "join", None
    #      [, lane, label]) | ("run_host", fn) entries; "run" entries between two collectives of a lane are recorded into ONE hipGraph.
    #      There is exactly one program per case:
    #        _plan_one_stream  one rank, everything on the caller's stream (overlap_critic=False: the per-kernel timing leg of bench.py, and
    #                          the form every other program must agree with);
    #        _plan_lanes       one rank (default): two lanes that never meet inside a step, ONE single-stream hipGraph each;
    #        _plan_dp          several ranks (or force_dp_plan): the same two lanes with the collectives between their graph segments.
    def _plan(self, batch: Dict[str, torch.Tensor], st: dict):
        m = self.loss_module
        if not m.critic_coef:
            raise NotImplementedError("PolicyUpdater expects the critic term (critic_coef > 0 in every TRPL config)")
        one_rank = m.world_size == 1 and not (self.group is not None and self.force_dp_plan)
        if one_rank and not self.overlap_critic:
            return self._plan_one_stream(batch, st)
        if one_rank:
            return self._plan_lanes(batch, st)
        return self._plan_dp(batch, st)

    def _critic_leaves(self):
        vf = self.loss_module.critic_network._network1
        ia, ib = vf.gnn.mlp_inner, vf.gnn.mlp_outer
        return (ia.lins[0].weight, ia.lins[0].bias, ia.norms[0].weight, ia.norms[0].bias, ia.lins[1].weight, ia.lins[1].bias,
                ib.lins[0].weight, ib.lins[0].bias, ib.norms[0].weight, ib.norms[0].bias, ib.lins[1].weight, ib.lins[1].bias,
                vf.final.weight, vf.final.bias)

    def _adam(self, st, lo, hi, i_, step_dev=None):
        """One optimizer's step over its slice [lo, hi) of the flat buffer (train.py:308-316); i_: 0 actor, 1 critic (clip workspace slot)."""
        coef = None
        if self.clip:  # train.py:308-310
            sq = st["zw"][23 + i_:24 + i_]
            coef = torch.empty(1, device=self.flat.device, dtype=torch.float32)
            hip.call("grl_clip_coef", self.gflat[lo:hi], hi - lo, float(self.max_norm), sq, coef)
        hip.call("grl_adam_step_dev", self.flat[lo:hi], self.gflat[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                 hi - lo, self.lr_dev, float(self.betas[0]), float(self.betas[1]), float(self.eps),
                 step_dev if step_dev is not None else self.step_dev, coef, 1.0)

    def _prep(self, batch, st, zero=None):
        """Inputs of a step: the batch with its variance diagonal, the two networks' observation lists, the fp64 workspace
        ((8 unused) | advantage sums (2) | loss sums (12) | maxes (2 x u32) | clip (2); every slot is WRITTEN by its producer).  ``zero``:
        the slice of the flat gradient THIS lane owns -- zeroed here when the folds accumulate (no overwrite mode)."""
        m = self.loss_module
        if not self._fold_overwrite and zero is not None:
            zero.zero_()
        b = dict(batch)
        if "var" not in b:
            b["var"] = b["covariance_matrix"].diagonal(dim1=-2, dim2=-1).contiguous()
        st["b"] = b
        st["obs"] = [b[k] for k in m.in_features]
        st["cobs"] = [b[k] for k in m.critic_in_features]
        st["zw"] = torch.empty(26, device=self.flat.device, dtype=torch.float64)

    def _actor_head(self, st, adv, adv_local):
        """Actor forward + fused loss kernel (actor terms only) + actor backward; -> the loss kernel's fold handle."""
        from .trpl import trpl_launch
        m, actor = self.loss_module, self.loss_module.actor_network
        zw = st["zw"]
        sums, maxes = zw[10:22], zw[22:23].view(torch.int32)
        ops.DEFERRED = []   # leaf-gradient folds of this backward are queued and executed by one launch at the lane's end
        loc, sigma = actor.forward_diag(*st["obs"], train=True)
        after_fwd = st.pop("after_forward", None)
        if after_fwd is not None:
            after_fwd()
        with torch.no_grad():
            fold_, _mx, dloc, dsigma, _ = trpl_launch(m, loc, sigma, None, st["b"], adv, sums=sums, maxes=maxes, defer_fold=True,
                                                      adv_local=adv_local)
        st.update(loc=loc.detach(), sigma=sigma.detach())
        # the lift's and the fiber basis' backward launches only feed the tail's fold: the first of the two waits for the other and they
        # share ONE launch (ops._tail_pre_offer)
        ops.TAIL_PRE = {} if ops.FUSE_TAIL_PRE else None
        try:
            torch.autograd.backward([loc, sigma], [dloc, dsigma])
            ops.flush_tail_pre()
        finally:
            ops.TAIL_PRE = None
        return fold_

    @staticmethod
    def _finish(st):
        """Host only: the output dict of views (recorded once, valid for every replay)."""
        a_loss, mt = st.pop("lv_main")
        mt = dict(mt)
        out = {"loss_objective": mt.pop("loss_objective_value"), "loss_critic": st.pop("c_loss"), "loc": st["loc"], "sigma": st["sigma"],
               "state_value": st["value"].unsqueeze(-1)}
        out.update(mt)
        st["out"] = out

    def _plan_one_stream(self, batch, st):
        """One rank, one stream: critic forward, actor forward, the fused loss kernel WITH the value terms, both backward passes, one fold,
        the optimizer step(s), reported values -- six closures on the caller's stream (one hipGraph when recorded)."""
        from .trpl import adv_stats_local, loss_values, trpl_launch
        m = self.loss_module
        actor, vf = m.actor_network, m.critic_network._network1
        leaves = self._critic_leaves()
        ow = self._fold_overwrite

        def s0():  # critic features, first critic stage, advantage statistics
            self._prep(batch, st, zero=self.gflat)
            with torch.no_grad():
                vf.train(True)
                _, x = vf.hyper_data.build_data(*st["cobs"], train=True)
                st["pipe"] = ops.DeepSetsPipeline(x, leaves, 1)
                st["pipe"].fwd1()
                st["adv"] = None
                if m.normalize_advantage and x.shape[0] > 1:
                    st["adv"] = st["zw"][8:10]
                    adv_stats_local(m, st["b"], st["adv"])

        def s1():
            st["pipe"].fwd2()

        def s2():  # value head, actor forward, fused TRPL kernel, actor backward, last critic stage backward
            ops.DEFERRED = []
            pipe = st["pipe"]
            value = pipe.fwd3()
            loc, sigma = actor.forward_diag(*st["obs"], train=True)
            with torch.no_grad():
                zw = st["zw"]
                sums, maxes, dloc, dsigma, dvalue = trpl_launch(m, loc, sigma, value, st["b"], st["adv"], sums=zw[10:22],
                                                                maxes=zw[22:23].view(torch.int32))
            torch.autograd.backward([loc, sigma], [dloc, dsigma])
            with torch.no_grad():
                pipe.bwd3(dvalue)
            st.update(loc=loc.detach(), sigma=sigma.detach(), value=value, sums=sums, maxes=maxes)

        def s3():
            st["pipe"].bwd2()

        def s4():
            with torch.no_grad():
                grads = st["pipe"].bwd1(leaves)
            assert all(g is None for g in grads), "critic parameters must own .grad views of the flat buffer"
            ops.flush_deferred_grads(overwrite=ow)
            ops.DEFERRED = None

        def s5():  # optimizers + reported values (train.py:308-316, trpl.py:280-321)
            with torch.no_grad():
                self.step_dev.add_(1)
                na, n = self.n_actor, self.flat.numel()
                # the two optimizers of train.py:120-127 have identical hyper-parameters and schedules: without per-network gradient
                # clipping their two Adam steps are ONE launch over the flat buffer (element-wise: the same numbers)
                for i_, (lo, hi) in enumerate(((0, na), (na, n)) if self.clip else ((0, n),)):
                    self._adam(st, lo, hi, i_)
                a_loss, c_loss, mt = loss_values(m, st["sums"], st["maxes"])
                st["lv_main"], st["c_loss"] = (a_loss, mt), c_loss

        return [("run", s0), ("run", s1), ("run", s2), ("run", s3), ("run", s4), ("run", s5), ("run_host", lambda: self._finish(st))]

    def _tail_args(self, lo, hi, cnt):
        return dict(grads=self.gflat[lo:hi], params=self.flat[lo:hi], exp_avg=self.exp_avg[lo:hi], exp_avg_sq=self.exp_avg_sq[lo:hi],
                    lr_dev=self.lr_dev, betas=self.betas, eps=self.eps, step_dev=cnt)

    def _plan_lanes(self, batch, st, cbatch=None, gate_in_graph=None, gate_override=None):
        """One rank as a two-lane PROGRAM of single-stream graphs.  This HIP runtime replays a captured graph with two branches through
        the host (hipGraphLaunch returned after 2/3 of the DEVICE time of a forked step; tools/ubench/graph_branches.py) and every
        cross-branch edge costs a 6-11 us gap; a graph boundary on a lane costs ~15 us as well.  So each lane is ONE graph and the lanes
        never meet inside a step (DESIGN.md findings 37, 38):
          actor's lane : features, lift, convolutions, read-out, fused loss kernel (actor terms only; the batch's advantage statistics are
                         summed inside it), backward, then ONE tail launch: fold + Adam over the actor's slice + reported values;
          critic's lane: features, the three forward stages, its OWN loss (clipped value loss: elementwise in the frame), the three
                         backward stages, tail: fold + Adam over the critic's slice.
        Actor and critic share no parameter and no intermediate (train.py:279-316 runs two backward passes and two optimizers); the lanes
        are forked at the step's start and joined at its end.  Each lane zeroes ITS OWN slice of the flat gradient when the folds
        accumulate (attention gate: torch's AccumulateGrad adds into .grad)."""
        from .trpl import report_dict, report_values, value_loss
        m = self.loss_module
        actor, vf = m.actor_network, m.critic_network._network1
        leaves = self._critic_leaves()
        ow = self._fold_overwrite
        na, n_all = self.n_actor, self.flat.numel()
        cb = cbatch if cbatch is not None else batch   # the critic lane's inputs (the epoch program gives each lane a private copy)
        # the gate as a launch INSIDE the critic's lane (grl_wait_flag_ge: capturable, target read from the lane's own device-side count) or
        # as a stream operation in front of its graph (hipStreamWaitValue32, round 5; GRL_GATE_STREAMWAIT=1)
        if gate_in_graph is None:
            gate_in_graph = self.gate_in_graph
        # the one-launch tail needs every leaf gradient of the lane in the fold queue (overwrite mode) and no clipping (which needs the
        # finished gradient norm before Adam)
        fuse_tail = not self.clip and ow
        # Gate the critic's lane behind the actor's first edge convolution?  It pays where the critic then finishes inside the actor's FORWARD
        # (its kernels cost the one-wave-per-SIMD backward launches far more than the forward ones: gated at the forward's END the step is 6 %
        # slower): measured on rigid HEPi +1.1 % at 4096 frames, +3 % at 512, +1 % at 32, 0 at 2048, -1.5 % at 1024, where a forward of
        # 0.35 ms is too short for it (profiles/r05_ab_critic_gate.txt, r05_ab_critic_gate_points.txt).  True (default) = that table.
        frames = next(int(v.shape[0]) for v in batch.values() if torch.is_tensor(v))
        mode = self.critic_after_first_conv
        gate = self._gate_for(frames) if gate_override is None else bool(gate_override)
        gate_point = mode if isinstance(mode, str) else "edge0"   # "edge0" | "fwd_end" (experiment: bench.py --critic-gate)

        def copy4(dst, src):   # dst[0] = src[0] (int32) on the current stream: one tiny launch
            import ctypes
            hip.call("grl_copy_many", (ctypes.c_void_p * 1)(dst.data_ptr()), (ctypes.c_void_p * 1)(src.data_ptr()), (ctypes.c_longlong * 1)(4), 1)

        def main_all():
            self._prep(batch, st, zero=self.gflat[:na])
            actor.hyper_data.bump_next = self.step_dev   # the step count rides on the lane's first launch (grl_build_features_bump)
            fired = []
            if gate and gate_point == "edge0":   # the critic's lane starts on this signal: flag = the step count, written right behind the first edge convolution
                def signal():
                    if actor.hyper_data.bump_next is not None:   # the step count has not been advanced yet (a calibrating pass in front of
                        return False                             # the step's own forward): not this edge convolution
                    fired.append(1)
                    # the signal rides on the NEXT launch of the lane (the fiber convolution behind this edge convolution writes the flag
                    # when it starts, ops.FiberConv): no 4-us copy launch on the step's chain
                    if ops.SIGNAL_IN_KERNEL:
                        ops.PENDING_SIGNAL = (self.lane_flag, self.step_dev)
                    else:
                        copy4(self.lane_flag, self.step_dev)
                    return True
                ops.AFTER_EDGE_HOOK = signal
            if gate and gate_point == "fwd_end":
                st["after_forward"] = lambda: copy4(self.lane_flag, self.step_dev)
            if gate and gate_point == "fiber0":   # experiment: the critic starts behind the first FIBER convolution (beside the ConvNeXt forward)
                def signal_f():
                    if actor.hyper_data.bump_next is not None:
                        return False
                    copy4(self.lane_flag, self.step_dev)
                    return True
                ops.AFTER_FIBER_HOOK = signal_f
            try:
                fold_ = self._actor_head(st, None, bool(m.normalize_advantage and st["obs"][0].shape[0] > 1))
            finally:
                ops.AFTER_EDGE_HOOK = None
                ops.AFTER_FIBER_HOOK = None
                unsent, ops.PENDING_SIGNAL = ops.PENDING_SIGNAL, None
            if unsent is not None:   # (no fiber convolution followed the edge convolution: send the signal by itself)
                copy4(*unsent)
            assert actor.hyper_data.bump_next is None, "the actor's feature launch did not take the step count"
            with torch.no_grad():
                done = False
                if fuse_tail:   # fold + Adam + reported values: ONE launch at the lane's end (ops.fold_adam_report)
                    o14 = torch.empty(14, device=self.flat.device, dtype=torch.float32)
                    ent = m.entropy_coef if m.entropy_bonus else 0.0
                    # (gated: the lane's closing signal -- see below -- rides on this launch)
                    done = ops.fold_adam_report(ow, self._tail_args(0, na, self.step_dev),
                                                dict(slots=fold_.slots, batch=fold_.batch, sums=fold_.sums, maxes=fold_.maxes,
                                                     ent_coef=ent, out14=o14), signal=(self.lane_flag, self.step_dev) if (gate and ops.SIGNAL_IN_KERNEL) else None)
                    if done:
                        a_loss, mt = report_dict(o14)
                if not done:
                    ops.flush_deferred_grads(overwrite=ow)
                    self._adam(st, 0, na, 0)
                    a_loss, _c, mt = report_values(m, fold_.slots, fold_.batch, fold_.sums, fold_.maxes)
                ops.DEFERRED = None
                if gate and not (done and ops.SIGNAL_IN_KERNEL):   # ... and once more at the lane's end, whatever happened above (an actor without an edge convolution; a
                    copy4(self.lane_flag, self.step_dev)   # signal that carried a stale count): the critic's lane can be late, it can never be stuck
                st.update(sums=fold_.sums, maxes=fold_.maxes, lv_main=(a_loss, mt))

        def critic_all():
            ops.DEFERRED = []
            with torch.no_grad():
                if self.critic_delay_us:   # (experiment knob, default 0: an idle one-wave kernel in front of the critic's lane)
                    hip.call("grl_calib_spin", int(self.critic_delay_us))
                if gate and gate_in_graph:   # until the actor's lane has signalled THIS step: flag >= the steps this lane has finished + 1
                    hip.call("grl_wait_flag_ge", self.lane_flag, self.step_dev_c, 1, 200000)
                pre = st.pop("critic_pre", None)
                if pre is not None:           # (epoch program: the lane's own minibatch gather)
                    pre()
                if not ow:
                    self.gflat[na:].zero_()   # on THIS lane, in front of its folds (never from the actor's lane)
                vf.train(True)
                # (inputs straight from the batch: this lane depends on nothing the actor's lane prepares, so it can be enqueued first)
                _, x = vf.hyper_data.build_data(*[cb[k] for k in m.critic_in_features], train=True, bump=self.step_dev_c)   # (+ the lane's step count)
                pipe = st["pipe"] = ops.DeepSetsPipeline(x, leaves, 1)
                pipe.fwd1()
                pipe.fwd2()
                value = st["value"] = pipe.fwd3()
                dvalue, c_loss, _ = value_loss(m, value, cb)
                pipe.bwd3(dvalue)
                pipe.bwd2()
                grads = pipe.bwd1(leaves)
                assert all(g is None for g in grads), "critic parameters must own .grad views of the flat buffer"
                if not (fuse_tail and ops.fold_adam_report(ow, self._tail_args(na, n_all, self.step_dev_c))):
                    ops.flush_deferred_grads(overwrite=ow)
                    self._adam(st, na, n_all, 1, self.step_dev_c)
                ops.DEFERRED = None
                st["c_loss"] = c_loss

        wait = [("wait_flag", None, "s", "critic_lane_start")] if (gate and not gate_in_graph) else []
        st["lanes"] = (main_all, critic_all)
        return [("fork", None), ("run", main_all), *wait, ("run", critic_all, "s"), ("join", None), ("run_host", lambda: self._finish(st))]

    def _plan_dp(self, batch, st):
        """Several ranks: the same two lanes, graph segments between the collectives (which stay eager torch.distributed calls).
          actor's lane : [features ... forward, fused loss kernel (actor terms), backward, fold, this rank's loss record] -> ONE all-reduce:
                         the ACTOR's slice of the flat gradient with the ranks' loss records riding in front of it -> [Adam, reported values];
                         the advantage statistics come from the rollout driver (one all-reduce per EPOCH, ``adv_stats`` column) or, for a
                         bare ``step(batch)``, from one more graph + collective at the head of the lane;
          critic's lane: its four LayerNorm-statistic reductions, its own loss, the all-reduce of ITS slice and of its loss sum, its Adam --
                         on a communicator of its own (``group_c``; collectives of one communicator execute in issue order on one internal
                         stream: a critic reduction waiting for a critic kernel must not hold back the actor's gradient all-reduce issued
                         behind it).  GRL_DP_ONE_COMM=1: both lanes on ``group`` (the documented fallback; same results).
        Every rank enqueues the SAME sequence of collectives per communicator, in the order of this list (host order = enqueue order);
        tests/test_dp_program_order.py checks that property of the program itself."""
        from .trpl import adv_stats_local, report_dict, value_loss
        m = self.loss_module
        world = m.world_size
        actor, vf = m.actor_network, m.critic_network._network1
        leaves = self._critic_leaves()
        ow = self._fold_overwrite
        na, n_all = self.n_actor, self.flat.numel()
        S = "s"
        # ``adv_stats`` in the batch ([B, 2] fp64, every row = the GLOBAL (sum, sum of squares) of this minibatch's advantages:
        # rollout.RolloutDriver.publish_advantage_stats): the statistics kernel, its all-reduce and the graph boundary behind it leave
        # the actor's lane -- two graphs and one collective on its path.
        published = m.normalize_advantage and "adv_stats" in batch

        def p_stats():
            self._prep(batch, st, zero=self.gflat[:na])
            st["adv"] = None
            if published:
                st["adv"] = batch["adv_stats"][0]
            elif m.normalize_advantage and st["obs"][0].shape[0] * world > 1:
                with torch.no_grad():
                    st["adv"] = st["zw"][8:10]
                    adv_stats_local(m, st["b"], st["adv"])

        # Large shards: the critic's lane waits (a launch of its own, grl_wait_flag_ge) until the actor's first edge convolution has finished --
        # beside that launch the critic's kernels cost it 60-170 us at 4096 frames (finding 42; DESIGN round 6).  The wait starts with the step
        # and ends inside the actor's forward: it is never resident during a backward launch.  GRL_DP_GATE_FROM frames (0 = never).
        frames_local = next((int(v.shape[0]) for v in batch.values() if torch.is_tensor(v)), 0)
        gate_dp = bool(self.dp_gate_from_frames) and self._work_frames(frames_local) >= self.dp_gate_from_frames   # (work-normalised, see _work_frames)

        def copy4(dst, src):
            import ctypes
            hip.call("grl_copy_many", (ctypes.c_void_p * 1)(dst.data_ptr()), (ctypes.c_void_p * 1)(src.data_ptr()), (ctypes.c_longlong * 1)(4), 1)

        def p_main():
            if published:
                p_stats()
            actor.hyper_data.bump_next = self.step_dev
            if gate_dp:
                def signal():
                    if actor.hyper_data.bump_next is not None:   # (a calibrating pass in front of the step's own forward: not this convolution)
                        return False
                    ops.PENDING_SIGNAL = (self.lane_flag, self.step_dev)   # rides on the fiber convolution behind the edge convolution
                    return True
                ops.AFTER_EDGE_HOOK = signal
            try:
                fold_ = self._actor_head(st, st["adv"], False)
            finally:
                ops.AFTER_EDGE_HOOK = None
                unsent, ops.PENDING_SIGNAL = ops.PENDING_SIGNAL, None
            if unsent is not None:
                copy4(*unsent)
            if gate_dp:   # ... and once more at the end of the segment: the critic's lane can be late, never stuck (its wait is bounded as well)
                copy4(self.lane_flag, self.step_dev)
            assert actor.hyper_data.bump_next is None
            st.update(sums=fold_.sums, maxes=fold_.maxes)
            with torch.no_grad():
                # this rank's loss sums / maxes as ONE record of float pairs in front of the flat gradient (own row, zeros in the
                # others): the SUM all-reduce of the actor's slice delivers every rank's record -- no collective of their own.  The
                # record and the fold of the slabs come from ONE launch (both only feed that all-reduce)
                if not ops.fold_record_pairs(ow, fold_.slots, fold_.batch, self.rank, world, self.gbuf[:self._rec]):
                    hip.call("grl_trpl_fold_record_pairs", fold_.slots, fold_.batch, self.rank, world, self.gbuf[:self._rec])
                    ops.flush_deferred_grads(overwrite=ow)
            ops.DEFERRED = None

        # (the reported values live in ONE buffer per recorded program: the tail may run as an eager launch behind the collective, below)
        o14 = torch.empty(14, device=self.flat.device, dtype=torch.float32)
        st["lv_main"] = report_dict(o14)

        def p_tail():   # behind the lane's collective: Adam on the reduced gradient, reported values of the delivered records
            with torch.no_grad():
                ent = m.entropy_coef if m.entropy_bonus else 0.0
                if self.clip:   # (the clip coefficient needs the reduced gradient's norm first: the separate launches)
                    self._adam(st, 0, na, 0)
                    hip.call("grl_trpl_report_record_pairs", self.gbuf[:self._rec], world, st["sums"], st["maxes"], float(ent), o14)
                else:           # Adam on the reduced slice + the reported values of the delivered records: ONE launch
                    hip.call("grl_adam_report_record_pairs", self.flat[:na], self.gflat[:na], self.exp_avg[:na], self.exp_avg_sq[:na], na,
                             self.lr_dev, float(self.betas[0]), float(self.betas[1]), float(self.eps), self.step_dev,
                             self.gbuf[:self._rec], world, st["sums"], st["maxes"], float(ent), o14)
        # ONE launch is cheaper issued eagerly than replayed as a one-node graph (a graph launch behind a collective costs ~13 us of lane time,
        # an eager kernel launch ~4): GRL_DP_EAGER_TAIL
        p_tail.eager = bool(self.dp_eager_tail and not self.clip)

        def q_fwd1():
            with torch.no_grad():
                if gate_dp:
                    hip.call("grl_wait_flag_ge", self.lane_flag, self.step_dev_c, 1, 200000)
                if not ow:
                    self.gflat[na:].zero_()
                vf.train(True)
                _, x = vf.hyper_data.build_data(*[batch[k] for k in m.critic_in_features], train=True, bump=self.step_dev_c)
                st["pipe"] = ops.DeepSetsPipeline(x, leaves, world)
                st["pipe"].fwd1()

        def q_fwd2():
            st["pipe"].fwd2()

        def q_fwd3():
            with torch.no_grad():
                st["value"] = st["pipe"].fwd3()
                dvalue, _mean, out2 = value_loss(m, st["value"], batch)
                st["vl"] = out2
                st["pipe"].bwd3(dvalue)

        def q_bwd2():
            st["pipe"].bwd2()

        def q_bwd1():
            keep = ops.DEFERRED
            ops.DEFERRED = []          # the critic's slabs are folded here, on its lane, into its slice of the flat gradient
            with torch.no_grad():
                grads = st["pipe"].bwd1(leaves)
                ops.flush_deferred_grads(overwrite=ow)
            ops.DEFERRED = keep
            assert all(g is None for g in grads), "critic parameters must own .grad views of the flat buffer"

        def q_tail():
            with torch.no_grad():
                self._adam(st, na, n_all, 1, self.step_dev_c)
                st["c_loss"] = st["vl"][1].float()   # (the all-reduced sum of the ranks' shares, already divided by B_global)

        # (host order = enqueue order: the critic's segments are interleaved so that its lane is fed early; each lane's own order is what
        #  the device sees.  p_stats comes first when present: it also prepares the step's inputs for both lanes.)
        head_ = [] if published else [("run", p_stats), ("sum", lambda: st["adv"], "m", "advantage_stats")]
        return [("fork", None), *head_,
                ("run", q_fwd1, S), ("sum", lambda: st["pipe"].stats1, S, "critic_ln1_fwd_stats"),
                ("run", q_fwd2, S), ("sum", lambda: st["pipe"].stats2, S, "critic_ln2_fwd_stats"),
                ("run", p_main),
                ("sum", lambda: self.gbuf[:self._rec + na], "m", "flat_gradient_actor+loss_records"),
                ("run", q_fwd3, S), ("sum", lambda: st["pipe"].bst2, S, "critic_ln2_bwd_stats"),
                ("run", q_bwd2, S), ("sum", lambda: st["pipe"].bst1, S, "critic_ln1_bwd_stats"),
                ("run", p_tail),
                ("run", q_bwd1, S), ("sum", lambda: self.gflat[na:], S, "flat_gradient_critic"), ("sum", lambda: st["vl"], S, "loss_critic_sum"),
                ("run", q_tail, S),
                ("join", None, "m", "join_critic_lane"), ("run_host", lambda: self._finish(st))]

    def program_outline(self, published: bool = True):
        """The data-parallel program as data, without running anything: [(kind, lane, label, communicator)] in HOST (enqueue) order, where
        communicator is "group" / "group_c" for collectives and None otherwise.  ``published``: the minibatch carries the epoch's advantage
        statistics (rollout.RolloutDriver.publish_advantage_stats) -- otherwise the lane starts with one more graph and collective.
        Needs no GPU (the closures are built, not called): tests/test_dp_program_order.py checks on four gloo ranks that every rank
        enqueues the same sequence per communicator and that the program has no dependency cycle across communicators -- the property the
        device-side RCCL run of two concurrently driven communicators rests on."""
        plan = self._plan_dp({"adv_stats": None} if published else {}, {})
        out = []
        for e in plan:
            kind, lane = e[0], (e[2] if len(e) > 2 else "m")
            label = e[3] if len(e) > 3 else None
            comm = None
            if kind == "sum":
                comm = "group_c" if (lane == "s" and self.group_c is not None) else "group"
            out.append((kind, lane, label, comm))
        return out

    def _critic_stream(self):
        if getattr(self, "_cstream", None) is None:
            # the LOWEST priority the device offers: the critic's small launches take the compute units the actor's kernels leave (heads,
            # tails, the latency-bound loss kernel) instead of displacing their workgroups (DESIGN.md finding 33)
            prio = 0
            if hasattr(torch.cuda.Stream, "priority_range"):
                try:
                    prio = max(torch.cuda.Stream.priority_range())
                except Exception:
                    prio = 0
            self._cstream = torch.cuda.Stream(priority=prio)
            self._cstream_priority = prio
        return self._cstream

    # ``collective_log``: None, or {} to record -- per label a list of (start event, end event, payload bytes) on the lane's stream: how long
    # the lane was held by each collective (wait for the other ranks + transfer).  Read with ``collective_summary()``.
    collective_log = None

    def _log_span(self, label, nbytes=0):
        """Context manager: HIP events on the current stream around a collective / cross-lane wait when the log is on."""
        import contextlib
        if self.collective_log is None or label is None:
            return contextlib.nullcontext()
        upd = self

        @contextlib.contextmanager
        def span():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            yield
            e1.record()
            upd.collective_log.setdefault(label, []).append((e0, e1, nbytes))
        return span()

    def collective_summary(self, n_steps: int):
        """{label: {per_step, mean_ms, max_ms, bytes}} from the recorded events (synchronises)."""
        torch.cuda.synchronize()
        out = {}
        for label, recs in (self.collective_log or {}).items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[label] = {"per_step": len(recs) / max(1, n_steps), "mean_ms": sum(ms) / len(ms), "max_ms": max(ms), "bytes": recs[0][2]}
        return out

    def _reduce(self, kind, t, label=None, lane="m"):
        """A collective of the program: "sum" all-reduce of ``t`` on the lane's communicator (the critic's lane: ``group_c``)."""
        import torch.distributed as dist
        if t is None:
            return
        if kind != "sum":
            raise ValueError(f"unknown program entry '{kind}'")
        group = self.group_c if (lane == "s" and self.group_c is not None) else self.group
        with self._log_span(label or kind, t.numel() * t.element_size()):
            if self._oneshot is not None and lane == "m" and t.data_ptr() == self._oneshot.payload.data_ptr() and t.numel() == self._oneshot.n:
                self._oneshot.all_reduce()   # (GRL_DP_ONESHOT=1: in place, on this lane's stream; every rank enqueues it at this point of the program)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    def _compile(self, batch):
        """Record the plan's segments into hipGraphs (adjacent segments without a reduction between them share one graph)."""
        m = self.loss_module
        # one sync, before anything is captured: the cached topology fits this minibatch
        m.actor_network.hyper_data.check_topology(*[batch[k] for k in m.in_features])
        m.critic_network._network1.hyper_data.check_topology(*[batch[k] for k in m.critic_in_features])
        self._static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        self._compile_one()

    def _compile_one(self):
        st = self._st = {}
        plan = self._plan(self._static, st)
        groups, cur, cur_lane = [], [], None
        for entry in plan:
            kind, item, lane = entry[0], entry[1], (entry[2] if len(entry) > 2 else "m")
            label = entry[3] if len(entry) > 3 else None
            if kind == "run" and getattr(item, "eager", False):   # a segment that is issued as plain launches at every step, not recorded
                if cur:
                    groups.append(("run", cur, cur_lane, None))
                    cur, cur_lane = [], None
                groups.append(("run_eager", item, lane, label))
                continue
            if kind == "run" and (not cur or lane == cur_lane):
                cur.append(item)
                cur_lane = lane
                continue
            if cur:
                groups.append(("run", cur, cur_lane, None))
                cur, cur_lane = [], None
            if kind == "run":
                cur, cur_lane = [item], lane
            elif kind == "run_host":   # host-only bookkeeping (output dict of views): once, when the step is recorded
                groups.append(("run_host_once", item, lane, label))
            else:
                groups.append((kind, item, lane, label))
        if cur:
            groups.append(("run", cur, cur_lane, None))
        program, pools = [], {}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        for kind, item, lane, label in groups:
            if kind == "run_host_once":
                item()
                continue
            if kind == "run_eager":
                program.append(("run", item, lane, label))
                continue
            if kind != "run":
                program.append((kind, item, lane, label))
                continue
            g = torch.cuda.CUDAGraph()
            # one allocator pool per lane: graphs of different lanes are replayed concurrently and must not share scratch memory
            # thread_local: background threads of the process (the collectives' watchdog) may keep issuing event queries
            with _no_gc_while_capturing(), torch.cuda.graph(g, pool=pools.get(lane), stream=side, capture_error_mode="thread_local"):
                for fn in item:
                    fn()
            pools[lane] = g.pool()
            program.append(("graph", g, lane, None))
        torch.cuda.current_stream().wait_stream(side)
        self._program = program

    def _refresh_static(self, batch):
        """Copy the minibatch into the static input buffers the recorded graphs read: one launch for all tensors."""
        import ctypes
        jobs = []
        for k, v in self._static.items():
            src = batch[k]
            if src is v:
                continue
            if src.shape != v.shape:
                raise ValueError(f"minibatch tensor '{k}' has shape {tuple(src.shape)}, the recorded step was captured for "
                                 f"{tuple(v.shape)}: call PolicyUpdater.reset_graph() before changing the minibatch size")
            if src.dtype != v.dtype or not src.is_contiguous() or src.device != v.device:
                v.copy_(src)  # host-resident / strided / other dtype: the ordinary path
            else:
                jobs.append((v.data_ptr(), src.data_ptr(), v.numel() * v.element_size()))
        for i in range(0, len(jobs), 24):
            part = jobs[i:i + 24]
            n = len(part)
            hip.call("grl_copy_many", (ctypes.c_void_p * n)(*[j[0] for j in part]), (ctypes.c_void_p * n)(*[j[1] for j in part]),
                     (ctypes.c_longlong * n)(*[j[2] for j in part]), n)

    def step_from(self, buf, idx: torch.Tensor) -> Dict[str, torch.Tensor]:
        """One update on the minibatch made of rows ``idx`` (device int64 [B]) of a device-resident ``rollout.RolloutBuffer``.
        With recorded graphs the rows are gathered by ONE launch straight into the static input buffers."""
        import ctypes
        keys = list(dict.fromkeys(list(self.loss_module.in_features) + list(self.loss_module.critic_in_features))) + ["action", "loc", "var" if "var" in buf.data else "covariance_matrix",
                                                     "sample_log_prob", "state_value", "advantage", "value_target"]
        if "adv_stats" in buf.data and self.group is not None:   # the epoch's published advantage statistics (rollout.RolloutDriver)
            keys.append("adv_stats")
        if not self.use_graph or self._program is None or int(idx.numel()) != self._static[keys[0]].shape[0]:
            return self.step(buf.rows(idx, keys))
        if any(k not in self._static for k in keys):   # (recorded without a key that is gathered now, e.g. adv_stats: re-record)
            return self.step(buf.rows(idx, keys))
        jobs = []
        for k in keys:
            dst, src = self._static[k], buf.flat(k)
            if dst.dtype != src.dtype or dst[0].numel() != src.shape[1]:
                return self.step(buf.rows(idx, keys))
            jobs.append((dst.data_ptr(), src.data_ptr(), src.shape[1] * src.element_size()))
        n = len(jobs)
        hip.call("grl_gather_rows_many", (ctypes.c_void_p * n)(*[j[0] for j in jobs]), (ctypes.c_void_p * n)(*[j[1] for j in jobs]),
                 (ctypes.c_longlong * n)(*[j[2] for j in jobs]), n, idx, int(idx.numel()))
        return self.step(self._static)

    # ---- several minibatch steps per launch (round 6).  A shard-sized step is ~35 dependent launches of 5-15 us; what it pays on top of
    #      them is the boundary of every replay (~13 us between two graph launches on a stream, ~7 us between the eager minibatch gather
    #      and the graph behind it) and, at 32 frames, the host's ~0.1 ms of enqueue work per step.  The minibatches of an epoch are known
    #      when it starts (train.py:258-261 iterates a sampler without replacement), so ``unroll`` consecutive steps are recorded into ONE
    #      graph per lane: the gathers ride inside (fixed rows of a static index matrix), each lane gathers ITS inputs into buffers of its
    #      own (the lanes share nothing, so no join between the steps of a launch), the critic's gate is a launch of its lane.
    def _epoch_ok(self) -> bool:
        m = self.loss_module
        return bool(self.use_graph and m.world_size == 1 and not (self.group is not None and self.force_dp_plan) and self.overlap_critic
                    and m.critic_coef and self.gate_in_graph and self.epoch_unroll > 1)

    def _lane_keys(self, buf):
        m = self.loss_module
        var = "var" if "var" in buf.data else "covariance_matrix"
        a = list(dict.fromkeys(list(m.in_features) + ["action", "loc", var, "sample_log_prob", "advantage"]))
        c = list(dict.fromkeys(list(m.critic_in_features) + ["state_value", "value_target"]))
        return a, c

    EPOCH_ROWS = 512   # index rows the cursor form keeps on the device (minibatches per load)

    def _compile_epoch(self, buf, idx0, U, cursor, gate=None):
        """Record ``U`` consecutive steps into one graph per lane.  ``cursor`` False: step j of a launch gathers the FIXED row j of a static
        [U, B] index matrix (filled per launch); True (U = 1): the launch gathers the row the lane's own device-side step count points at in
        a static [EPOCH_ROWS, B] matrix loaded once per call -- nothing but graph launches per step."""
        import ctypes
        m = self.loss_module
        B = int(idx0.numel())
        ka, kc = self._lane_keys(buf)
        sa, sc = buf.rows(idx0, ka), buf.rows(idx0, kc)      # static inputs, one private set per lane
        m.actor_network.hyper_data.check_topology(*[sa[k] for k in m.in_features])
        m.critic_network._network1.hyper_data.check_topology(*[sc[k] for k in m.critic_in_features])
        rows = self.EPOCH_ROWS if cursor else U
        idx_static = torch.zeros(rows, B, device=idx0.device, dtype=torch.int64)
        base_a = torch.zeros(1, device=idx0.device, dtype=torch.int32)
        base_c = torch.zeros(1, device=idx0.device, dtype=torch.int32)

        def gather_args(static, keys):
            n = len(keys)
            for k in keys:
                if static[k].dtype != buf.flat(k).dtype or static[k][0].numel() != buf.flat(k).shape[1]:
                    raise RuntimeError(f"rollout tensor '{k}' cannot be gathered row-wise into the recorded step's inputs")
            return ((ctypes.c_void_p * n)(*[static[k].data_ptr() for k in keys]), (ctypes.c_void_p * n)(*[buf.flat(k).data_ptr() for k in keys]),
                    (ctypes.c_longlong * n)(*[buf.flat(k).shape[1] * buf.flat(k).element_size() for k in keys]), n)
        ga, gc = gather_args(sa, ka), gather_args(sc, kc)

        def gather(args, j, count, base):
            if cursor:
                return lambda: hip.call("grl_gather_rows_many_cur", *args, idx_static, B, count, base, rows)
            row = idx_static[j]
            return lambda: hip.call("grl_gather_rows_many", *args, row, B)
        sts, mains, critics = [], [], []
        for j in range(U):
            st = {}
            self._plan_lanes(sa, st, cbatch=sc, gate_in_graph=True, gate_override=gate)
            main_all, critic_all = st.pop("lanes")
            mains.append((gather(ga, j, self.step_dev, base_a), main_all))
            critics.append((st, gather(gc, j, self.step_dev_c, base_c), critic_all))
            sts.append(st)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ga_graph, gc_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with _no_gc_while_capturing(), torch.cuda.graph(ga_graph, stream=side, capture_error_mode="thread_local"):
            for g_, main_all in mains:
                g_()
                main_all()
        with _no_gc_while_capturing(), torch.cuda.graph(gc_graph, stream=side, capture_error_mode="thread_local"):
            for st, g_, critic_all in critics:
                st["critic_pre"] = g_              # (behind the lane's gate, in front of its features: critic_all runs it)
                critic_all()
        torch.cuda.current_stream().wait_stream(side)
        for st in sts:
            self._finish(st)
        tail = [] if cursor else [("join", None, "m", None)]   # (cursor form: the lanes are joined once, when the call returns)
        self._epoch = dict(key=(B, U, cursor, id(buf)), idx=idx_static, base=(base_a, base_c), sts=sts, keep=(sa, sc, ga, gc, buf),
                           program=[("fork", None, "m", None), ("graph", ga_graph, "m", None), ("graph", gc_graph, "s", None)] + tail)

    # The size thresholds of the lane policy (_gate_for, run_minibatches) were measured on rigid_insertion_multi HEPi, whose compact graph has
    # this many edges per frame (kNN + task edges); every other workload is placed on that scale by ITS edge count -- the edge kernels are
    # 60 % of every step, so a cloth minibatch of 512 frames is "a rigid minibatch of ~1900 frames" to the policy, not a small one
    POLICY_EDGES_PER_FRAME = 64

    def _work_frames(self, frames: int) -> int:
        """``frames`` of this workload on the scale the lane policy was measured on: the cached topology's edges / POLICY_EDGES_PER_FRAME
        (the topology of a size exists once its first, eager step has run; before that the frame count itself)."""
        hd = getattr(self.loss_module.actor_network, "hyper_data", None)
        topo = hd._cache.get(frames) if hd is not None else None
        if not topo or not topo.get("edges"):
            return frames
        return max(1, round(sum(es.n_edges for es in topo["edges"].values()) / self.POLICY_EDGES_PER_FRAME))

    def _gate_for(self, frames: int) -> bool:
        """Is the critic's lane gated behind the actor's first edge convolution at this minibatch size?  (see _plan_lanes)"""
        mode = self.critic_after_first_conv
        return bool(mode) and (isinstance(mode, str) or not (768 <= self._work_frames(frames) < 2048))

    def run_minibatches(self, buf, idx_rows: torch.Tensor, unroll: Optional[int] = None):
        """The updates of consecutive minibatches: ``idx_rows`` [M, B] int64 (device), row j = the rollout rows of minibatch j (what
        ``rollout.RolloutDriver.epoch_minibatches`` hands out).  One rank with recorded lanes takes one of two recorded forms:
          * ``unroll`` steps per launch where the critic's lane is not gated, or the launches are smaller than the chip (<= 64 frames):
            no boundary at all between the steps of a launch (-3 % at 32 frames, -1.5 % at 1024);
          * one step per launch with the gathers inside (index row picked by the lane's device-side step count) where the lane IS gated:
            a gate waiting inside a multi-step launch is a resident wave during the previous step's one-wave-per-SIMD backward kernels,
            which then find 255 free compute units for 256 workgroups (+10 % at 512 frames, +20 % at 4096: DESIGN.md, round 6) -- here
            the lanes are forked per step so that the gate starts with the step, and joined once at the end.
        Otherwise, and for the first (eager) step of a size and the remainder, a loop of ``step_from``.  Returns the loss dict of the last
        step; ``self.last_outs`` holds the dicts of the last launch's steps."""
        M, B = int(idx_rows.shape[0]), int(idx_rows.shape[1])
        U = int(unroll or self.epoch_unroll)
        out, j = None, 0
        if not self._epoch_ok() or U <= 1:
            for j in range(M):
                out = self.step_from(buf, idx_rows[j])
            return out
        while j < M and B not in getattr(self, "_eager_sizes", ()):     # the first step of a size runs eagerly (topology, calibration, checks)
            out = self.step_from(buf, idx_rows[j])
            j += 1
        # Which form?  Measured on rigid HEPi (profiles/r06_ab_ungated_unroll.txt, r06_ab_unroll.txt): several steps per launch with the critic's
        # lane UNGATED (it runs ahead inside the launch; the lanes share nothing) beats the gated per-step program up to 2048 frames
        # (-3.7 % at 128, -1.2 % at 256 / 512, -2 % at 2048) and loses to it at 4096 (+1.7 %), where the critic's 0.25 ms of kernels beside
        # the wrong launches cost more than the boundaries of a 3 ms step; with the gate INSIDE a multi-step launch it loses everywhere
        # above 64 frames (a resident waiting wave during the previous step's one-wave-per-SIMD backward kernels).
        Bw = self._work_frames(B)   # (the thresholds are on the scale of the workload they were measured on)
        gate_here = self._gate_for(B) and (Bw <= self.epoch_unroll_max_gated_frames or Bw >= self.epoch_gated_from_frames)
        gated_big = gate_here and Bw > self.epoch_unroll_max_gated_frames
        cursor = gated_big and self.epoch_cursor
        # That table is ONE workload's; on the others the better of the two forms differs by 1-2 % either way (cloth and rope
        # prefer the gated per-step program at 512 frames, the two-agent EMPN the multi-step launch: profiles/r06_ab_policy.txt).  Both forms
        # give bitwise the same update, so above 64 work-frames the choice is MEASURED once per size on the running program (_tune_form:
        # alternating blocks of U steps of each, HIP events) whenever a call brings enough minibatches; until then the table decides.
        form = self.form_by_size.get(B)
        if (form is None and self.autotune_form and not self.epoch_cursor and Bw > self.epoch_unroll_max_gated_frames
                and M - j >= self.tune_minibatches(U)):
            j, out = self._tune_form(buf, idx_rows, j, U)
            form = self.form_by_size[B]
        per_step = (form == "per_step") if form is not None else (gated_big and not cursor)
        if form == "unrolled":
            cursor, gate_here = False, False   # (the measured multi-step launch is the ungated one)
        if cursor:
            U = 1
        elif per_step:     # a gate inside a multi-step launch is a resident wave during the previous step's backward: the per-step program
            while j < M:
                out = self.step_from(buf, idx_rows[j])
                j += 1
            return out
        if M - j >= U:
            if self._epoch is None or self._epoch["key"] != (B, U, cursor, id(buf)):
                self.loss_module._global_steps = self.steps
                self._compile_epoch(buf, idx_rows[j], U, cursor, gate=gate_here)
            ep = self._epoch
            if cursor:
                main = torch.cuda.current_stream()
                while j < M:
                    n = min(M - j, self.EPOCH_ROWS)
                    main.wait_stream(self._critic_stream())          # every earlier step of BOTH lanes is behind us: the counts are final
                    ep["idx"][:n].copy_(idx_rows[j:j + n])
                    ep["base"][0].copy_(self.step_dev)
                    ep["base"][1].copy_(self.step_dev_c)
                    for _ in range(n):
                        self.steps += 1
                        try:
                            self._execute(ep["program"])
                        except BaseException:
                            self.steps -= 1
                            raise
                    j += n
                main.wait_stream(self._critic_stream())
                self.last_outs = [ep["sts"][0]["out"]]
                out = self.last_outs[-1]
            while not cursor and M - j >= U:
                ep["idx"].copy_(idx_rows[j:j + U])
                self.steps += U
                try:
                    self._execute(ep["program"])
                except BaseException:
                    self.steps -= U
                    raise
                j += U
                self.last_outs = [st["out"] for st in ep["sts"]]
                out = self.last_outs[-1]
        while j < M:
            out = self.step_from(buf, idx_rows[j])
            j += 1
        return out

    TUNE_ROUNDS = 2   # alternating blocks per form in _tune_form

    def tune_minibatches(self, U: Optional[int] = None) -> int:
        """Minibatches one measurement of the two recorded forms consumes (they are ordinary updates, in order)."""
        U = int(U or self.epoch_unroll)
        return U + 3 + 2 * self.TUNE_ROUNDS * U

    def _tune_form(self, buf, idx_rows, j, U):
        """Measure, on the running program, which recorded form is faster at this minibatch size: ``U`` steps per launch with the critic's lane
        ungated, or one step per launch with the lane gated as ``_gate_for`` says.  Both are recorded and replayed once, then TUNE_ROUNDS
        alternating blocks of U steps of each are timed with HIP events on the caller's stream (both lanes joined at every block boundary);
        ONE host synchronisation at the end.  Every step is an ordinary update of the next minibatch, and the two forms produce bitwise the
        same update (tests/test_gpu_rollout.py), so the measurement changes nothing but the time.  -> (next j, last loss dict)."""
        B = int(idx_rows.shape[1])
        main, cs = torch.cuda.current_stream(), self._critic_stream()
        out = None

        def unrolled(rows):
            if self._epoch is None or self._epoch["key"] != (B, U, False, id(buf)):
                self.loss_module._global_steps = self.steps
                self._compile_epoch(buf, rows[0], U, False, gate=False)
            ep = self._epoch
            ep["idx"].copy_(rows)
            self.steps += U
            try:
                self._execute(ep["program"])
            except BaseException:
                self.steps -= U
                raise
            self.last_outs = [st["out"] for st in ep["sts"]]
            return self.last_outs[-1]

        def per_step(rows):
            o = None
            for r in rows:
                o = self.step_from(buf, r)
            return o

        out = unrolled(idx_rows[j:j + U]); j += U          # records (and runs) the multi-step launch
        out = per_step(idx_rows[j:j + 3]); j += 3          # records the per-step program, first replays
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * self.TUNE_ROUNDS + 1)]
        main.wait_stream(cs)
        ev[0].record(main)
        for r in range(self.TUNE_ROUNDS):
            out = unrolled(idx_rows[j:j + U]); j += U
            main.wait_stream(cs)
            ev[2 * r + 1].record(main)
            out = per_step(idx_rows[j:j + U]); j += U
            main.wait_stream(cs)
            ev[2 * r + 2].record(main)
        ev[-1].synchronize()
        t_u = min(ev[2 * r].elapsed_time(ev[2 * r + 1]) for r in range(self.TUNE_ROUNDS)) / U
        t_p = min(ev[2 * r + 1].elapsed_time(ev[2 * r + 2]) for r in range(self.TUNE_ROUNDS)) / U
        self.form_by_size[B] = "unrolled" if t_u <= t_p else "per_step"
        self.form_times[B] = {"unrolled_ms_per_step": t_u, "per_step_ms_per_step": t_p, "steps_per_block": U, "blocks_per_form": self.TUNE_ROUNDS}
        return j, out

    def reset_graph(self):
        """Drop the recorded step (next step re-records): needed when the minibatch size changes."""
        self._program, self._static, self._epoch = None, None, None
        self._eager_sizes = set()
        self.form_by_size, self.form_times = {}, {}

    def _check_calibrated(self):
        """Data parallel: the first training forward of a fresh actor re-initialises the conv kernels from rank-local data
        (conv.py:104-105).  Let it happen once, on every rank, BEFORE the first update, then adopt rank 0's result -- otherwise each
        replica would rescale its own weights (views of ``flat``) and the replicas would diverge for good."""
        # (the latch FIRST: ``hasattr(gnn, "calibrated")`` evaluates the property, which reads the per-conv flags from the device -- until
        #  round 4 every data-parallel step paid a device synchronisation here and the host never ran ahead of the device)
        if self.group is None or getattr(self, "_calib_synced", False):
            return None
        actor = self.loss_module.actor_network
        gnn = getattr(actor, "gnn", None)
        if gnn is None or not hasattr(gnn, "calibrated"):
            return None
        self._calib_synced = True
        return not gnn.calibrated   # True: this rank's actor still has to calibrate -> the caller syncs afterwards

    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        self.loss_module._global_steps = self.steps
        need_sync = self._check_calibrated()
        if need_sync:
            actor, m = self.loss_module.actor_network, self.loss_module
            with torch.no_grad():
                actor.forward_diag(*[batch[k] for k in m.in_features], train=True)   # calibrates on this rank's shard
            self.sync_replicas()                                                       # ... and rank 0's factors win everywhere
        self.steps += 1
        try:
            return self._step(batch)
        except BaseException:
            # the critic lane's gate waits until the DEVICE-side step count (advanced by the actor's feature launch) reaches the HOST's: a
            # step that raised before that launch was enqueued must not leave the host one ahead for good -- the next step's wait would
            # never be satisfied and the process would hang on the device instead of raising.  After a failure the host count
            # is at most the device's: the gate may then release early, it can never be stuck.
            self.steps -= 1
            raise

    def _step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        B = next(v.shape[0] for v in batch.values() if torch.is_tensor(v))
        seen = getattr(self, "_eager_sizes", None)
        if seen is None:
            seen = self._eager_sizes = set()
        if not self.use_graph or B not in seen:     # the first step of a minibatch size always runs eagerly: it builds the
            seen.add(B)                             # cached topology of that size and the kernels' one-time attributes
            st = {}
            hooks, torch_fed = [], []
            if not self._overwrite_checked:
                # overwrite mode (no zeroing launch) is only right while NO leaf gradient arrives through torch's AccumulateGrad, which would
                # add into a never-zeroed .grad: a hook on a leaf fires exactly for such a gradient (ops' backward functions hand None to
                # autograd for the leaves whose slabs they queue).  Checked once, on the first eager step.
                hooks = [p.register_hook(lambda g, i=i: torch_fed.append(i) if g is not None else None) for i, p in enumerate(self.params)]   # (autograd calls a leaf hook with None when a backward function returned no gradient for it)
            try:
                self._execute([(e[0], e[1], e[2] if len(e) > 2 else "m", e[3] if len(e) > 3 else None) for e in self._plan(batch, st)])
            finally:
                for h in hooks:
                    h.remove()
            if hooks:
                self._overwrite_checked = True
                if torch_fed:
                    raise RuntimeError(f"{len(set(torch_fed))} parameter(s) received their gradient from torch autograd instead of the fold queue "
                                       "while PolicyUpdater runs in overwrite mode (their .grad is never zeroed): this model must be added to "
                                       "the exceptions of PolicyUpdater._fold_overwrite")
            return st["out"]
        if self._program is not None and (any(self._static[k].shape != batch[k].shape for k in self._static if k in batch)
                                          or ("adv_stats" in batch) != ("adv_stats" in self._static)):
            self.reset_graph()                      # another minibatch size (or the published advantage statistics appeared /
            self._eager_sizes.add(B)                # disappeared: a different data-parallel plan): record again for it
        if self._program is None:
            try:
                self._compile(batch)
            except Exception as e:
                torch.cuda.synchronize()
                if not self.allow_eager_fallback:
                    raise RuntimeError(
                        f"hipGraph capture of the policy-update step failed ({type(e).__name__}: {e}).  Pass use_graph=False, or "
                        "allow_eager_fallback=True to continue with eager launches (several times slower for small minibatches).") from e
                import sys
                print(f"[geometry_rl_amd] hipGraph capture failed ({type(e).__name__}: {e}); continuing with eager launches "
                      "(allow_eager_fallback=True)", file=sys.stderr)
                self.use_graph, self._program, self._static, self.mode = False, None, None, "eager (graph capture failed)"
                return self._step(batch)
        self._refresh_static(batch)
        self._execute(self._program)
        return self._st["out"]

    def _execute(self, program):
        """Run a program: ("run" closure | "graph" replay | "sum" collective | "fork" | "join", item, lane, label).  Lane "m" is the
        caller's stream, lane "s" the critic's stream; "fork": the side lane waits for the main lane, "join": the reverse."""
        main = torch.cuda.current_stream()
        side = None
        for kind, item, lane, label in program:
            if kind in ("fork", "join"):
                side = side or self._critic_stream()
                if kind == "join":
                    with self._log_span(label):   # how long the main lane stood waiting for the critic's lane
                        main.wait_stream(side)
                else:
                    side.wait_stream(main)
                continue
            if lane == "s":
                side = side or self._critic_stream()
                with torch.cuda.stream(side):
                    self._do(kind, item, label, lane)
            else:
                self._do(kind, item, label)

    def _do(self, kind, item, label=None, lane="m"):
        if kind == "wait_flag":   # (current stream = the critic's) until the actor's lane has signalled THIS step (host count = device count)
            hip.stream_wait_value32(self.lane_flag, self.steps)
            return
        if kind in ("run", "run_host"):
            item()
        elif kind == "graph":
            item.replay()
        else:
            self._reduce(kind, item() if item is not None else None, label, lane)


def gae(reward, done, terminated, values, gamma=0.99, lmbda=0.95):
    """Shifted GAE (train.py:134-140,249-251): reward/done/terminated [N,T], values [N,T+1] -> advantage, value_target [N,T]."""
    hip.check_f32(reward, values)
    N, T = reward.shape
    adv = torch.empty_like(reward)
    tgt = torch.empty_like(reward)
    hip.call("grl_gae_scan", reward.contiguous(), done.to(torch.uint8).contiguous(), terminated.to(torch.uint8).contiguous(),
             values.contiguous(), adv, tgt, N, T, float(gamma), float(lmbda))
    return adv, tgt


# ---------------------------------------------------------------------------------------------------- checkpoints
# train.py:336-368 saves {"env", "actor": actor.state_dict(), "critic": critic.state_dict(), "reward"} where ``actor`` is torchrl's
# ProbabilisticActor(TensorDictModule(policy)) -- a TensorDictSequential whose first entry wraps the policy -- and ``critic`` is
# ValueOperator(BaseCritic) (utils_algo_graph.py:146-158,200-203).  The wrappers only add key prefixes.
ACTOR_PREFIX = "module.0.module."
CRITIC_PREFIX = "module."


def _strip(sd, prefix):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


def load_reference_checkpoint(ckpt, actor, critic=None, strict=True, trust=False):
    """Load a reference ``model_checkpoint_*.pth`` (path or the loaded dict) into the HIP-backed actor / critic (play.py:194-205).
    Parameter names are identical (PyG ModuleDict key mangling and the ``callibrated`` buffers included); returns ckpt["reward"].

    The file is read with ``weights_only=True`` (tensors and plain containers only).  The reference also pickles ``env.state_dict()``
    into the same file (train.py:343-351), which may hold arbitrary objects: if the safe load fails, pass ``trust=True`` to fall back
    to a full unpickle -- only for files you produced yourself, unpickling executes code.

    Note on parity of a loaded policy: the KL covariance projection used when training continues here is validated against this
    repository's KKT restatement (ITPAL's source is not in the reference checkout), see DESIGN.md section 2."""
    if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
        try:
            ckpt = torch.load(ckpt, map_location="cpu", weights_only=True)
        except Exception as e:
            if not trust:
                raise RuntimeError(f"{ckpt!r} cannot be read with weights_only=True ({type(e).__name__}: {e}); pass trust=True to "
                                   "unpickle it fully (executes code from the file)") from e
            ckpt = torch.load(ckpt, map_location="cpu", weights_only=False)
    actor.load_state_dict(_strip(ckpt["actor"], ACTOR_PREFIX), strict=strict)
    if critic is not None and "critic" in ckpt:
        critic.load_state_dict(_strip(ckpt["critic"], CRITIC_PREFIX), strict=strict)
    return ckpt.get("reward")


def reference_checkpoint(actor, critic, reward=0.0, env_state=None):
    """The dict train.py:343-351 writes, so the reference's play.py can load a policy trained here."""
    return {"env": env_state if env_state is not None else {},
            "actor": {ACTOR_PREFIX + k: v.detach().cpu().clone() for k, v in actor.state_dict().items()},
            "critic": {CRITIC_PREFIX + k: v.detach().cpu().clone() for k, v in critic.state_dict().items()},
            "reward": reward}
