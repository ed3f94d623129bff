"""Wiring of actor, critic, projection and loss for one task config -- the counterpart of
``examples/torchrl/builders/agent.py:14-80`` + ``builders/utils_algo_graph.py:208-276`` without Hydra / the Isaac env object
(observation layout comes from a TaskSpec instead of ``env.observation_manager``), plus the policy-update driver that
replaces the inner loop of ``examples/torchrl/train.py:258-316``."""
from dataclasses import dataclass
from typing import Dict, Optional

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


class PolicyUpdater:
    """One policy-update step = loss forward, actor + critic backward, optional clip_grad_norm_ per network, two Adam(lr,
    eps=1e-5) steps (train.py:279-316).  Parameters of both networks live in ONE flat fp32 buffer (gradients likewise), so a
    data-parallel run needs a single RCCL all-reduce of the gradient per step and Adam is a single kernel per optimizer.

    The step is laid out as an explicit plan (``_plan``) of device-only segments, each free of host synchronisation, so with
    ``use_graph=True`` they are recorded once into hipGraphs (torch.cuda.CUDAGraph) and replayed -- the ~3 ms of per-step launch
    overhead disappears, which is what strong scaling over 8 GPUs needs (512 frames per GPU are < 1 ms of device time).
      * one rank: a single graph; the critic's small kernels run on a second stream beside the actor's (fork / join inside the graph);
      * several ranks: a two-lane program (``_execute``): the main lane carries the actor, the TRPL kernel, the gradient folding and
        Adam, the side lane the critic together with its four statistic all-reduces; segments are separate graphs and the
        collectives stay ordinary eager torch.distributed calls between the replays.  The main lane waits for one collective per
        step, the all-reduce of the flat gradient."""

    def __init__(self, loss_module: TRPLLoss, lr=3e-4, eps=1e-5, betas=(0.9, 0.999), clip_grad_norm=False, max_grad_norm=1.0,
                 group=None, use_graph=False, overlap_critic=True, allow_eager_fallback=False):
        self.loss_module, self.group = loss_module, group
        self.overlap_critic = overlap_critic and os.environ.get("GRL_OVERLAP_CRITIC", "1") != "0"   # one rank only: critic kernels on a second stream beside the actor's
        self.overlap_folds = overlap_critic and os.environ.get("GRL_OVERLAP_FOLDS", "0") != "0"   # the leaf-gradient folds on a third
        # stream, one launch per backward op (ops.FOLD_STREAM).  Measured round 3 and left OFF: 0.858 vs 0.783 ms per 512-frame step, 3.42 vs
        # 3.38 ms at 4096 frames -- ten small launches with cross-stream edges in the graph cost more than the one 45 us launch they replace
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
        self.step_dev_c = torch.zeros(1, device=dev, dtype=torch.int32)  # the same count kept by the critic's lane (one rank, two lanes)
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
        self._pending = []   # asynchronous collectives in flight
        # Leaf gradients are WRITTEN by the one fold launch at the end of the backward pass (ops.flush_deferred_grads(overwrite=True)) instead
        # of accumulated into a zeroed buffer: the per-step ``gflat.zero_()`` launch -- the first node of the recorded step, in front of the
        # fork of the two lanes -- disappears.  Valid while every leaf gradient of the step comes through the fold queue: not with the stock
        # torch transformer actor or the attention gate (torch's AccumulateGrad adds into .grad), nor with per-op folds on a third stream.
        gnn = getattr(loss_module.actor_network, "gnn", None)
        self._fold_overwrite = (os.environ.get("GRL_FOLD_OVERWRITE", "1") != "0" and not self.overlap_folds
                                and not getattr(loss_module.actor_network, "post_fc", False)
                                and not any(getattr(mod, "attention", False) for mod in (gnn.modules() if gnn is not None else [])))
        # one rank: the critic folds and applies its own gradients on its lane (its own Adam launch over its slice of the flat buffer): the
        # actor's lane neither carries them nor waits for the critic before its fold
        self._critic_own_adam = os.environ.get("GRL_CRITIC_OWN_ADAM", "1") != "0"
        # one rank: the step as a two-lane program of single-stream graphs (default) instead of ONE graph with a fork / join inside -- see _plan
        self._lanes = os.environ.get("GRL_LANES", "1") != "0"
        # the critic's lane reduces on a communicator of its own (see _plan): created collectively, here, in construction order
        self.group_c = None
        if group is not None and os.environ.get("GRL_DP_JOINED", "0") == "0":
            import torch.distributed as dist
            try:
                self.group_c = dist.new_group(ranks=dist.get_process_group_ranks(group))
            except Exception:
                self.group_c = None
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
            self._program = None   # recorded launches carry the old scalar

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

    # ---- the plan: [("run", fn) | ("sum", tensor getter) | ("max", tensor getter)] --------------------------------------
    def _plan(self, batch: Dict[str, torch.Tensor], st: dict):
        from . import ops
        from .trpl import adv_stats_local, head_launch, loss_values, report_dict, report_values, trpl_launch, value_loss
        m = self.loss_module
        world = m.world_size
        # GRL_FORCE_DP_PLAN=1 with a process group of ONE rank: the data-parallel program (lanes with joins, graph segments between the
        # collectives, every all-reduce issued) on one GPU -- what a shard's step costs before any inter-GPU latency (bench.py --dp-plan)
        one_rank = world == 1 and not (self.group is not None and os.environ.get("GRL_FORCE_DP_PLAN", "0") != "0")
        actor = m.actor_network
        vf = m.critic_network._network1
        ia, ib = vf.gnn.mlp_inner, vf.gnn.mlp_outer
        leaves = (ia.lins[0].weight, ia.lins[0].bias, ia.norms[0].weight, ia.norms[0].bias, ia.lins[1].weight, ia.lins[1].bias,
                  ib.lins[0].weight, ib.lins[0].bias, ib.norms[0].weight, ib.norms[0].bias, ib.lins[1].weight, ib.lins[1].bias,
                  vf.final.weight, vf.final.bias)
        if not m.critic_coef:
            raise NotImplementedError("PolicyUpdater expects the critic term (critic_coef > 0 in every TRPL config)")

        ow = self._fold_overwrite

        def adam(lo, hi, i_, step_dev=None):   # one optimizer's step over its slice of the flat buffer (train.py:308-316)
            coef = None
            if self.clip:  # train.py:308-310
                sq = st["zw"][23 + i_:24 + i_]
                coef = torch.empty(1, device=self.flat.device, dtype=torch.float32)
                hip.call("grl_clip_coef", self.gflat[lo:hi], hi - lo, float(self.max_norm), sq, coef)
            hip.call("grl_adam_step_dev", self.flat[lo:hi], self.gflat[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                     hi - lo, self.lr_dev, float(self.betas[0]), float(self.betas[1]), float(self.eps),
                     step_dev if step_dev is not None else self.step_dev, coef, 1.0)

        def s0():  # critic features, first critic stage, advantage statistics
            if not ow:
                self.gflat.zero_()
            b = dict(batch)
            if "var" not in b:
                b["var"] = b["covariance_matrix"].diagonal(dim1=-2, dim2=-1).contiguous()
            st["b"] = b
            st["obs"] = [b[k] for k in m.in_features]
            st["cobs"] = [b[k] for k in m.critic_in_features]
            with torch.no_grad():
                vf.train(True)
                _, x = vf.hyper_data.build_data(*st["cobs"], train=True)
                # one fp64 workspace per step: (8 unused) | advantage sums (2) | loss sums (12) | maxes (2 x u32) | clip (2); every slot is
                # WRITTEN by its producer (ABI 203): no zeroing launch
                zw = st["zw"] = torch.empty(26, device=x.device, dtype=torch.float64)
                st["pipe"] = ops.DeepSetsPipeline(x, leaves, world)
                st["pipe"].fwd1()
                st["adv"] = None
                if m.normalize_advantage and x.shape[0] * world > 1:
                    st["adv"] = zw[8:10]
                    adv_stats_local(m, b, st["adv"])

        def s1():
            st["pipe"].fwd2()

        def s2():  # value head, actor forward, fused TRPL kernel, actor backward, last critic stage backward
            ops.DEFERRED = []   # leaf-gradient folds of this backward are queued and executed by one launch in s4
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
                if not st.pop("step_bumped", False):
                    self.step_dev.add_(1)
                na, n = self.n_actor, self.flat.numel()
                if st.pop("critic_adam_done", False):
                    adam(0, na, 0)   # the critic's optimizer has run on its own lane
                else:
                    # the two optimizers of train.py:120-127 have identical hyper-parameters and schedules: without per-network gradient
                    # clipping their two Adam steps are ONE launch over the flat buffer (element-wise: the same numbers)
                    for i_, (lo, hi) in enumerate(((0, na), (na, n)) if self.clip else ((0, n),)):
                        adam(lo, hi, i_)
                join = st.pop("join_side", None)
                if join is not None:   # the critic's lane ends here (nothing of the actor's lane is queued behind this: no cost on its path)
                    join()
                a_loss, c_loss, mt = st.pop("lv", None) or loss_values(m, st["sums"], st["maxes"])
                out = {"loss_objective": mt.pop("loss_objective_value"), "loss_critic": c_loss, "loc": st["loc"], "sigma": st["sigma"],
                       "state_value": st["value"].unsqueeze(-1)}
                out.update(mt)
                st["out"] = out

        # ---- one rank: no reduction separates the critic stages, so the whole critic (small, latency-bound launches) runs on
        #      a second stream beside the actor -- forward beside the actor forward, backward beside the actor backward -- and
        #      fills the SIMDs the big kernels leave idle at their heads and tails.  Recorded into the graph as a fork / join.
        def o_fwd():
            b = dict(batch)
            if "var" not in b:
                b["var"] = b["covariance_matrix"].diagonal(dim1=-2, dim2=-1).contiguous()
            st["b"] = b
            st["obs"] = [b[k] for k in m.in_features]
            st["cobs"] = [b[k] for k in m.critic_in_features]
            cur, cs = torch.cuda.current_stream(), self._critic_stream()
            if not ow:
                self.gflat.zero_()
            cs.wait_stream(cur)
            with torch.cuda.stream(cs), torch.no_grad():
                # inputs of the fused loss kernel and of Adam that depend on the minibatch alone come FIRST on this lane: the step's workspace
                # (every slot written by its producer: no zeroing), the advantage statistics, the optimizer step count -- the critic's
                # forward behind them is what the actor's first edge convolution may have to wait for
                zw = st["zw"] = torch.empty(26, device=self.flat.device, dtype=torch.float64)
                st["adv"] = None
                if m.normalize_advantage and st["obs"][0].shape[0] > 1:
                    st["adv"] = zw[8:10]
                    adv_stats_local(m, b, st["adv"])
                self.step_dev.add_(1)
                st["step_bumped"] = True
                vf.train(True)
                _, x = vf.hyper_data.build_data(*st["cobs"], train=True)
                pipe = st["pipe"] = ops.DeepSetsPipeline(x, leaves, 1)
                pipe.fwd1()
                ev1 = torch.cuda.Event()
                ev1.record(cs)
                pipe.fwd2()
                value = pipe.fwd3()
            ops.DEFERRED = []
            if self.overlap_folds:   # leaf-gradient folds beside the backward kernels, on a third stream (ops.FOLD_STREAM)
                ops.FOLD_STREAM = self._fold_stream()
                ops.FOLD_STREAM.wait_stream(cur)   # behind the zeroing of the flat gradient
            # The MFMA kernels size their grids to fill every CU exactly (all LDS, all VGPRs): a critic workgroup still resident when
            # one of them starts displaces one of ITS workgroups, which then runs as a second round behind the others -- the first edge
            # convolution of the EMPN step took 751 instead of 566 us that way (profiles/r03_stream_kernels_ab.txt).  So when the critic's forward
            # is short enough to hide behind the actor's prologue (gather, features, lift: streaming kernels that share CUs gracefully)
            # the first edge convolution waits for it; a long critic forward (cloth: 239 rows per frame, 0.7 ms) keeps running beside
            # the actor instead -- waiting would cost more than the displacement.
            mode = os.environ.get("GRL_CRITIC_JOIN", "auto")
            rows = int(x.shape[0]) * (int(x.shape[1]) if x.dim() == 3 else 1)
            if mode != "0":
                ops.PRE_EDGE_HOOK = lambda n_nodes: (cur.wait_stream(cs) if (mode == "1" or (mode == "auto" and rows <= 4 * n_nodes)) else
                                                     cur.wait_event(ev1) if mode in ("fwd1", "auto") else None)
            try:
                loc, sigma = actor.forward_diag(*st["obs"], train=True)
            finally:
                ops.PRE_EDGE_HOOK = None   # (one-shot; never left behind for another caller's edge convolution)
            cur.wait_stream(cs)   # join: the fused loss kernel needs the values
            with torch.no_grad():   # (the fold of the per-workgroup loss sums is deferred: reported values only, off the actor's lane)
                fold, maxes, dloc, dsigma, dvalue = trpl_launch(m, loc, sigma, value, b, st["adv"], sums=zw[10:22],
                                                                maxes=zw[22:23].view(torch.int32), defer_fold=True)
            cs.wait_stream(cur)   # fork: critic backward beside the actor backward
            own = self._critic_own_adam and not self.overlap_folds
            with torch.cuda.stream(cs), torch.no_grad():
                sums, maxes = fold()
                st["lv"] = loss_values(m, sums, maxes)   # reported values only: beside the backward pass, not behind Adam
                pipe.bwd3(dvalue)
                pipe.bwd2()
                grads = pipe.bwd1(leaves)
                if own:   # the critic's gradients are complete: folded and applied HERE, on its lane (only its slabs are queued so far)
                    na, n = self.n_actor, self.flat.numel()
                    ops.flush_deferred_grads(overwrite=ow)
                    adam(na, n, 1)
                    st["critic_adam_done"] = True
            assert all(g is None for g in grads), "critic parameters must own .grad views of the flat buffer"
            torch.autograd.backward([loc, sigma], [dloc, dsigma])
            if own:
                st["join_side"] = lambda: cur.wait_stream(cs)   # joined behind the actor's Adam (s5): off the actor's path
            else:
                cur.wait_stream(cs)   # join: every partial slab is complete
            ops.flush_deferred_grads(overwrite=ow)
            ops.DEFERRED = None
            ops.FOLD_STREAM = None
            st.update(loc=loc.detach(), sigma=sigma.detach(), value=value, sums=sums, maxes=maxes)

        if one_rank and self.overlap_critic and not self._lanes:
            return [("run", o_fwd), ("run", s5)]
        if one_rank and not self.overlap_critic:   # one rank, one stream
            return [("run", s0), ("run", s1), ("run", s2), ("run", s3), ("run", s4), ("run", s5)]

        # ---- several ranks: two lanes.  The critic with ALL FOUR of its reductions runs on a second stream ("s" items) beside the
        #      actor forward / backward; the loss sums / maxes (reported values only) are reduced asynchronously behind the actor
        #      backward.  The main lane waits for ONE collective per step: the gradient all-reduce.  Tensors that cross lanes stay
        #      referenced in ``st`` for the whole step, so neither allocator pool can hand their memory out while the other lane
        #      still uses it.
        def m_prep(zero=None):
            if not ow:
                (self.gflat if zero is None else zero).zero_()
            b = dict(batch)
            if "var" not in b:
                b["var"] = b["covariance_matrix"].diagonal(dim1=-2, dim2=-1).contiguous()
            st["b"] = b
            st["obs"] = [b[k] for k in m.in_features]
            st["cobs"] = [b[k] for k in m.critic_in_features]
            st["zw"] = torch.empty(26, device=self.flat.device, dtype=torch.float64)   # every slot is written by its producer

        def c_fwd1():
            with torch.no_grad():
                vf.train(True)
                _, x = vf.hyper_data.build_data(*st["cobs"], train=True)
                st["pipe"] = ops.DeepSetsPipeline(x, leaves, world)
                st["pipe"].fwd1()
                st["adv"] = None
                if m.normalize_advantage and x.shape[0] * world > 1:
                    st["adv"] = st["zw"][8:10]
                    adv_stats_local(m, st["b"], st["adv"])

        def c_fwd3():
            st["value"] = st["pipe"].fwd3()

        def a_fwd():
            ops.DEFERRED = []   # leaf-gradient folds of both backward passes are queued and executed by one launch in ``fold``
            st["loc_g"], st["sigma_g"] = actor.forward_diag(*st["obs"], train=True)

        def head():  # fused TRPL kernel
            loc, sigma = st["loc_g"], st["sigma_g"]
            defer = False
            with torch.no_grad():
                zw = st["zw"]
                sums, maxes, dloc, dsigma, dvalue = trpl_launch(m, loc, sigma, st["value"], st["b"], st["adv"], sums=zw[10:22],
                                                                maxes=zw[22:23].view(torch.int32), defer_fold=defer)
            if defer:
                st["fold"], sums = sums, None
            st.update(loc=loc.detach(), sigma=sigma.detach(), sums=sums, maxes=maxes, dloc=dloc, dsigma=dsigma, dvalue=dvalue)

        def c_bwd3():
            with torch.no_grad():
                st["pipe"].bwd3(st["dvalue"])

        def c_bwd1():
            with torch.no_grad():
                grads = st["pipe"].bwd1(leaves)
            assert all(g is None for g in grads), "critic parameters must own .grad views of the flat buffer"

        def a_bwd():
            torch.autograd.backward([st.pop("loc_g"), st.pop("sigma_g")], [st["dloc"], st["dsigma"]])

        # read-out forward, loss kernel and read-out backward as ONE launch (trpl.head_launch, GRL_FUSED_HEAD=1).  Built as VERDICT r3 item 1c
        # asked, measured, and left OFF: 0.3533 -> 0.3577 / 0.7218 -> 0.7278 / 3.3634 -> 3.3624 ms per step at 32 / 512 / 4096 frames on one
        # box -- the three launches' 8 + 12 + 12 us are dependent latency that a workgroup pays between its barriers just the same; what
        # the fusion saves (two ~2 us launch gaps) the 16-wave fold of the decoder's gradient sums spends (tests/test_gpu_fused_head.py
        # keeps both paths equal).
        fused_head = (os.environ.get("GRL_FUSED_HEAD", "0") != "0" and not getattr(actor, "post_fc", False)
                      and hasattr(actor, "latent_diag") and hasattr(getattr(actor, "gnn", None), "decoder"))

        def a_head(adv, adv_local):
            """forward + loss (actor terms) + backward of the actor; -> the loss kernel's fold handle"""
            zw = st["zw"]
            sums, maxes = zw[10:22], zw[22:23].view(torch.int32)
            B_ = st["obs"][0].shape[0]
            if fused_head and st["b"]["action"].reshape(B_, -1).shape[1] <= 16:   # (the loss kernel's widest instance: 16 lanes per frame)
                ops.DEFERRED = []
                lat = actor.latent_diag(*st["obs"], train=True)
                with torch.no_grad():
                    fold_, loc, sigma, dlat = head_launch(m, actor, lat, st["b"], adv, sums, maxes, adv_local=adv_local)
                st.update(loc=loc, sigma=sigma)
                torch.autograd.backward([lat], [dlat])
                return fold_
            a_fwd()
            loc, sigma = st["loc_g"], st["sigma_g"]
            with torch.no_grad():
                fold_, _mx, dloc, dsigma, _ = trpl_launch(m, loc, sigma, None, st["b"], adv, sums=sums, maxes=maxes, defer_fold=True,
                                                          adv_local=adv_local)
            st.update(loc=loc.detach(), sigma=sigma.detach(), dloc=dloc, dsigma=dsigma)
            a_bwd()
            return fold_

        def fold():
            ops.flush_deferred_grads(overwrite=ow)
            ops.DEFERRED = None

        S = "s"
        if one_rank and self.overlap_critic and self._lanes:
            # One rank as a two-lane PROGRAM of single-stream graphs: this HIP runtime replays a captured graph with two branches through
            # the host -- hipGraphLaunch returned after 2/3 of the DEVICE time of the step (0.26 / 0.48 / 2.1 ms at 32 / 512 / 4096 frames
            # against 24 us for a one-stream graph of the same kernels; tools/ubench/graph_branches.py: two independent 20-kernel chains
            # replay in 139 us, host-bound, one 40-kernel chain in 77) -- and every cross-branch edge costs a 6-11 us gap.  A graph boundary
            # on a lane costs ~15 us as well (the next graph's launch), so each lane is ONE graph and the lanes never meet inside a step:
            #   actor's lane : features, lift, convolutions, read-out, fused loss kernel (actor terms only; the batch's advantage statistics
            #                  are summed inside it), backward, fold, Adam over the actor's slice, fold + evaluation of the reported values;
            #   critic's lane: features, the three forward stages, its OWN loss (clipped value loss: elementwise in the frame), the three
            #                  backward stages, fold, Adam over the critic's slice.
            # Actor and critic share no parameter and no intermediate (train.py:279-316 runs two backward passes and two optimizers); the
            # lanes are forked at the step's start and joined at its end.
            # (clipping needs the finished gradient norm first; gradients that reach .grad through torch's AccumulateGrad -- attention gate,
            # stock transformer: exactly the cases without overwrite mode -- are not in the fold queue, so their parameters would be skipped)
            fuse_tail = os.environ.get("GRL_FUSED_TAIL", "1") != "0" and not self.clip and ow
            adam_args = lambda lo, hi, cnt: dict(grads=self.gflat[lo:hi], params=self.flat[lo:hi], exp_avg=self.exp_avg[lo:hi],
                                                 exp_avg_sq=self.exp_avg_sq[lo:hi], lr_dev=self.lr_dev, betas=self.betas, eps=self.eps,
                                                 step_dev=cnt)

            def main_all():
                m_prep()
                st["step_bumped"] = True
                actor.hyper_data.bump_next = self.step_dev   # the step count rides on the lane's first launch (grl_build_features_bump)
                fold_ = a_head(None, bool(m.normalize_advantage and st["obs"][0].shape[0] > 1))
                assert actor.hyper_data.bump_next is None, "the actor's feature launch did not take the step count"
                with torch.no_grad():
                    done = False
                    if fuse_tail:   # fold + Adam + reported values: ONE launch at the lane's end (ops.fold_adam_report)
                        o14 = torch.empty(14, device=self.flat.device, dtype=torch.float32)
                        ent = m.entropy_coef if m.entropy_bonus else 0.0
                        done = ops.fold_adam_report(ow, adam_args(0, self.n_actor, self.step_dev),
                                                    dict(slots=fold_.slots, batch=fold_.batch, sums=fold_.sums, maxes=fold_.maxes,
                                                         ent_coef=ent, out14=o14))
                        if done:
                            ops.DEFERRED = None
                            a_loss, mt = report_dict(o14)
                    if not done:
                        fold()
                        adam(0, self.n_actor, 0)
                        a_loss, _c, mt = report_values(m, fold_.slots, fold_.batch, fold_.sums, fold_.maxes)
                    st.update(sums=fold_.sums, maxes=fold_.maxes, lv_main=(a_loss, mt))

            def critic_all():
                ops.DEFERRED = []
                with torch.no_grad():
                    vf.train(True)
                    _, x = vf.hyper_data.build_data(*st["cobs"], train=True, bump=self.step_dev_c)   # (+ the lane's step count)
                    pipe = st["pipe"] = ops.DeepSetsPipeline(x, leaves, 1)
                    pipe.fwd1()
                    pipe.fwd2()
                    value = st["value"] = pipe.fwd3()
                    dvalue, c_loss, _ = value_loss(m, value, st["b"])
                    pipe.bwd3(dvalue)
                    pipe.bwd2()
                    grads = pipe.bwd1(leaves)
                    assert all(g is None for g in grads), "critic parameters must own .grad views of the flat buffer"
                    if not (fuse_tail and ops.fold_adam_report(ow, adam_args(self.n_actor, self.flat.numel(), self.step_dev_c))):
                        ops.flush_deferred_grads(overwrite=ow)
                        adam(self.n_actor, self.flat.numel(), 1, self.step_dev_c)
                    ops.DEFERRED = None
                    st["c_loss"] = c_loss

            def finish():
                a_loss, mt = st.pop("lv_main")
                mt = dict(mt)
                out = {"loss_objective": mt.pop("loss_objective_value"), "loss_critic": st.pop("c_loss"), "loc": st["loc"], "sigma": st["sigma"],
                       "state_value": st["value"].unsqueeze(-1)}
                out.update(mt)
                st["out"] = out

            # (finish only builds the dict of output views: it records nothing; kept as a "run" so that eager steps execute it too)
            return [("fork", None), ("run", main_all), ("run", critic_all, S), ("join", None), ("run_host", finish)]
        # ---- several ranks, round 4: the lanes do not meet inside a step here either.  The critic's lane carries ALL of the critic -- its four
        #      LayerNorm-statistic reductions, its own loss (value_loss: elementwise in the frame), the all-reduce of ITS slice of the flat
        #      gradient and its optimizer step -- on a communicator of its own (collectives of one communicator execute in issue order on
        #      one internal stream: a critic reduction that waits for a critic kernel would hold back the actor's gradient all-reduce
        #      issued behind it).  The actor's lane: statistics of the advantages (all-reduced: 16 bytes) | forward, fused loss kernel
        #      (actor terms), backward, fold | all-reduce of the actor's slice, all-gather of the ranks' loss records (one collective
        #      for sums and maxes) | Adam, reported values: THREE graphs and three collectives on its path (round 3: five graphs, two
        #      joins with the critic's lane, one synchronous + two asynchronous collectives and a wait).
        #      (collectives carry a label as fourth entry: PolicyUpdater.collective_log / bench.py's N > 1 line report them by name)
        if os.environ.get("GRL_DP_JOINED", "0") == "0":
            na, n_all = self.n_actor, self.flat.numel()
            # ``adv_stats`` in the batch ([B, 2] fp64, every row = the GLOBAL (sum, sum of squares) of this minibatch's advantages:
            # rollout.RolloutDriver.publish_advantage_stats, one all-reduce per EPOCH): the statistics kernel, its all-reduce and the graph
            # boundary behind it leave the actor's lane -- two graphs and two collectives on its path.
            published = m.normalize_advantage and "adv_stats" in batch

            def p_stats():   # (each lane zeroes its own slice of the flat gradient when the folds accumulate)
                m_prep(self.gflat[:na])
                st["adv"] = None
                if published:
                    st["adv"] = batch["adv_stats"][0]
                elif m.normalize_advantage and st["obs"][0].shape[0] * world > 1:
                    with torch.no_grad():
                        st["adv"] = st["zw"][8:10]
                        adv_stats_local(m, st["b"], st["adv"])

            def p_main():
                if published:
                    p_stats()
                st["step_bumped"] = True
                actor.hyper_data.bump_next = self.step_dev
                fold_ = a_head(st["adv"], False)
                assert actor.hyper_data.bump_next is None
                with torch.no_grad():
                    # this rank's loss sums / maxes as ONE record of float pairs in front of the flat gradient (own row, zeros in the
                    # others): the SUM all-reduce of the actor's slice delivers every rank's record -- no collective of their own
                    hip.call("grl_trpl_fold_record_pairs", fold_.slots, fold_.batch, self.rank, world, self.gbuf[:self._rec])
                st.update(sums=fold_.sums, maxes=fold_.maxes)
                fold()

            def p_tail():   # behind the two collectives of the lane: Adam on the reduced gradient, reported values of the gathered records
                with torch.no_grad():
                    adam(0, na, 0)
                    ent = m.entropy_coef if m.entropy_bonus else 0.0
                    o14 = torch.empty(14, device=self.flat.device, dtype=torch.float32)
                    hip.call("grl_trpl_report_record_pairs", self.gbuf[:self._rec], world, st["sums"], st["maxes"], float(ent), o14)
                    a_loss, mt = report_dict(o14)
                    st["lv_main"] = (a_loss, None, mt)

            def q_fwd1():
                with torch.no_grad():
                    if not ow:
                        self.gflat[na:].zero_()
                    vf.train(True)
                    _, x = vf.hyper_data.build_data(*[batch[k] for k in m.critic_in_features], train=True, bump=self.step_dev_c)
                    st["pipe"] = ops.DeepSetsPipeline(x, leaves, world)
                    st["pipe"].fwd1()

            def q_fwd3():
                with torch.no_grad():
                    st["value"] = st["pipe"].fwd3()
                    dvalue, _mean, out2 = value_loss(m, st["value"], batch)
                    st["vl"] = out2
                    st["pipe"].bwd3(dvalue)

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
                    adam(na, n_all, 1, self.step_dev_c)
                    st["c_loss"] = st["vl"][1].float()   # (the all-reduced sum of the ranks' shares, already divided by B_global)

            def finish_dp():
                a_loss, _c, mt = st.pop("lv_main")
                mt = dict(mt)
                out = {"loss_objective": mt.pop("loss_objective_value"), "loss_critic": st.pop("c_loss"), "loc": st["loc"], "sigma": st["sigma"],
                       "state_value": st["value"].unsqueeze(-1)}
                out.update(mt)
                st["out"] = out

            # (host order = enqueue order: the critic's segments are interleaved so that its lane is fed early; each lane's own order is what
            #  the device sees.  p_stats comes first: it also prepares the step's inputs for both lanes.)
            head_ = [] if published else [("run", p_stats), ("sum", lambda: st["adv"], "m", "advantage_stats")]
            return [("fork", None), *head_,
                    ("run", q_fwd1, S), ("sum", lambda: st["pipe"].stats1, S, "critic_ln1_fwd_stats"),
                    ("run", s1, S), ("sum", lambda: st["pipe"].stats2, S, "critic_ln2_fwd_stats"),
                    ("run", p_main),
                    ("sum", lambda: self.gbuf[:self._rec + na], "m", "flat_gradient_actor+loss_records"),
                    ("run", q_fwd3, S), ("sum", lambda: st["pipe"].bst2, S, "critic_ln2_bwd_stats"),
                    ("run", s3, S), ("sum", lambda: st["pipe"].bst1, S, "critic_ln1_bwd_stats"),
                    ("run", p_tail),
                    ("run", q_bwd1, S), ("sum", lambda: self.gflat[na:], S, "flat_gradient_critic"), ("sum", lambda: st["vl"], S, "loss_critic_sum"),
                    ("run", q_tail, S),
                    ("join", None, "m", "join_critic_lane"), ("run_host", finish_dp)]

        # ---- the joined form of rounds 2-3 (GRL_DP_JOINED=1): the critic's lane is forked after the input preparation and after the
        #      fused loss kernel and joined in front of both
        plan = [("run", m_prep), ("fork", None),
                ("run", c_fwd1, S), ("sum", lambda: st["pipe"].stats1, S, "critic_ln1_fwd_stats"), ("sum", lambda: st["adv"], S, "advantage_stats"),
                ("run", a_fwd),
                ("run", s1, S), ("sum", lambda: st["pipe"].stats2, S, "critic_ln2_fwd_stats"), ("run", c_fwd3, S),
                ("join", None, "m", "join_critic_forward"),
                ("run", head), ("fork", None),
                ("run", c_bwd3, S), ("sum", lambda: st["pipe"].bst2, S, "critic_ln2_bwd_stats"), ("run", s3, S),
                ("sum", lambda: st["pipe"].bst1, S, "critic_ln1_bwd_stats"),
                ("run", c_bwd1, S),
                ("sum_async", lambda: st["sums"], "m", "loss_sums"), ("max_async", lambda: st["maxes"], "m", "loss_maxes"),   # loss terms: already scaled by 1/B_global
                ("run", a_bwd),
                ("join", None, "m", "join_critic_backward"), ("wait", None, "m", "wait_async_loss_terms"),
                ("run", fold), ("sum", lambda: self.gflat, "m", "flat_gradient")]
        plan += [("run", s5)]
        return plan

    def _fold_stream(self):
        if getattr(self, "_fstream", None) is None:
            self._fstream = torch.cuda.Stream()
        return self._fstream

    def _critic_stream(self):
        if getattr(self, "_cstream", None) is None:
            # the LOWEST priority the device offers: the critic's small launches take the compute units the actor's kernels leave (heads,
            # tails, the latency-bound loss kernel) instead of displacing their workgroups (DESIGN.md finding 33)
            prio = 0
            if os.environ.get("GRL_CRITIC_PRIO", "1") != "0" and hasattr(torch.cuda.Stream, "priority_range"):
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
        import torch.distributed as dist
        group = self.group_c if (lane == "s" and self.group_c is not None) else self.group
        if kind == "wait":
            with self._log_span(label):
                for w in self._pending:
                    w.wait()
            self._pending = []
            return
        if t is None:
            return
        if kind == "gather":   # t = (out [world, n], in [n])
            out_t, in_t = t
            with self._log_span(label or kind, in_t.numel() * in_t.element_size()):
                if dist.get_backend(group) == "nccl":
                    dist.all_gather_into_tensor(out_t, in_t, group=group)
                else:   # (gloo: the list form; rows of out_t are contiguous views)
                    dist.all_gather(list(out_t.unbind(0)), in_t, group=group)
            return
        nbytes = t.numel() * t.element_size()
        if kind in ("sum_async", "max_async"):
            op = dist.ReduceOp.SUM if kind == "sum_async" else dist.ReduceOp.MAX
            with self._log_span((label or kind) + " (issue only: asynchronous)", nbytes):
                self._pending.append(dist.all_reduce(t, op=op, group=group, async_op=True))
            return
        with self._log_span(label or kind, nbytes):
            dist.all_reduce(t, op=dist.ReduceOp.SUM if kind == "sum" else dist.ReduceOp.MAX, group=group)

    def _compile(self, batch):
        """Record the plan's segments into hipGraphs (adjacent segments without a reduction between them share one graph)."""
        m = self.loss_module
        # one sync, before anything is captured: the cached topology fits this minibatch
        m.actor_network.hyper_data.check_topology(*[batch[k] for k in m.in_features])
        m.critic_network._network1.hyper_data.check_topology(*[batch[k] for k in m.critic_in_features])
        self._static = {k: v.clone() for k, v in batch.items() if torch.is_tensor(v)}
        st = self._st = {}
        plan = self._plan(self._static, st)
        groups, cur, cur_lane = [], [], None
        for entry in plan:
            kind, item, lane = entry[0], entry[1], (entry[2] if len(entry) > 2 else "m")
            label = entry[3] if len(entry) > 3 else None
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
            if kind != "run":
                program.append((kind, item, lane, label))
                continue
            g = torch.cuda.CUDAGraph()
            # one allocator pool per lane: graphs of different lanes are replayed concurrently and must not share scratch memory
            # thread_local: background threads of the process (the collectives' watchdog) may keep issuing event queries
            with torch.cuda.graph(g, pool=pools.get(lane), stream=side, capture_error_mode="thread_local"):
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

    def reset_graph(self):
        """Drop the recorded step (next step re-records): needed when the minibatch size changes."""
        self._program, self._static = None, None
        self._eager_sizes = set()

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
        B = next(v.shape[0] for v in batch.values() if torch.is_tensor(v))
        seen = getattr(self, "_eager_sizes", None)
        if seen is None:
            seen = self._eager_sizes = set()
        if not self.use_graph or B not in seen:     # the first step of a minibatch size always runs eagerly: it builds the
            seen.add(B)                             # cached topology of that size and the kernels' one-time attributes
            st = {}
            self._execute([(e[0], e[1], e[2] if len(e) > 2 else "m", e[3] if len(e) > 3 else None) for e in self._plan(batch, st)])
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
                self.steps -= 1
                return self.step(batch)
        self._refresh_static(batch)
        self._execute(self._program)
        return self._st["out"]

    def _actor_stream(self):
        """One rank, two lanes: the actor's lane runs on a stream of its own with the HIGHEST priority the device offers (the critic's lane
        keeps the default): when both lanes have workgroups to place, the actor's go first, and the critic's small launches take what the
        actor's kernels leave -- their heads and tails, the latency-bound loss kernel -- instead of sitting on compute units the first edge
        convolution then finds occupied (DESIGN.md finding 33: 319 -> 382 us for that launch at 4096 frames with both lanes at one priority)."""
        if getattr(self, "_astream", None) is None:
            prio = 0
            # MEASURED round 4 (tools/prio_ab.sh, one box) and left OFF: 32 / 512 frames 0.370 -> 0.409 / 0.705 -> 0.736 ms per step, 4096
            # frames 3.195 -> 3.196, cloth 5.42 -> 5.48, EMPN 4.80 -> 4.91: the two extra stream hand-overs per step (caller -> own stream
            # -> caller) cost more than the priority buys -- the critic's resident workgroups are not evicted by it
            if os.environ.get("GRL_ACTOR_PRIO", "0") != "0" and hasattr(torch.cuda.Stream, "priority_range"):
                try:
                    prio = min(torch.cuda.Stream.priority_range())
                except Exception:
                    prio = 0
            self._astream = torch.cuda.Stream(priority=prio) if prio != 0 else False
            self._astream_priority = prio
        return self._astream or None

    def _execute(self, program):
        """Run a program: ("run" closure | "graph" replay | collective | "fork" | "join" | "wait", item, lane).  Lane "m" is the
        caller's stream (one rank with two lanes: a high-priority stream of the updater's, joined back into the caller's at the end), lane
        "s" the critic stream; "fork": the side lane waits for the main lane, "join": the reverse."""
        caller = torch.cuda.current_stream()
        own = self._actor_stream() if (self._lanes and self.group is None and self.overlap_critic) else None   # (off by default)
        main = own or caller
        if own is not None:
            own.wait_stream(caller)
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
            elif own is not None:
                with torch.cuda.stream(own):
                    self._do(kind, item, label)
            else:
                self._do(kind, item, label)
        if own is not None:
            caller.wait_stream(own)

    def _do(self, kind, item, label=None, lane="m"):
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
