"""Batched merge environment on PyTorch tensors over the C ABI (host side of the hot path).

`VecMergeEnv` is the product class: E independent episodes x N CAVs live as a struct of arrays in
HBM and every reset/step is ONE call into libmm_hip.so (hand-written HIP, gfx950).  It refuses
to run without that library and a GPU -- there is no CPU fallback.

`BatchedMergeEnv` is the device-agnostic plumbing (state views, output buffers, ctypes calls); the
tests also drive it with the CPU oracle library on host tensors to check parity.

Mirrors the reference's Env API batched over E: reset -> (obs[E,N,n_s], avail[E,N,5]),
step(actions[E,N]) -> (obs, reward[E], done[E], info{...tensors...}).
"""
import ctypes as C
import os

import torch

from . import _cabi as abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# MM_HIP_LIB: kernel-tuning experiments load an alternative build of the SAME library (never the oracle)
HIP_LIB = os.environ.get("MM_HIP_LIB") or os.path.join(_HERE, "csrc", "libmm_hip.so")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedMergeEnv(object):
    n_a = 5  # merge_env_v1.py:27

    def __init__(self, clib, E, N, env_id="merge-multi-agent-v1", config=None, device="cpu",
                 cbf_eta=0.0, cbf_tau=None, auto_reset=False, obs_f64=False, seed=0, first_env=0,
                 trace=False, debug_flags=0, n_hdv=0, qp_solver="ipm", draw_counts=False, num_cav=0, skip_outputs=()):
        self.clib, self.E, self.N = clib, int(E), int(N)
        self.env_id = env_id
        self.device = torch.device(device)
        self.config = abi.default_env_config(env_id)
        if config:
            self.config.update(config)
        self.cbf_eta, self.cbf_tau = cbf_eta, cbf_tau
        self.auto_reset, self.obs_f64, self.seed = auto_reset, obs_f64, seed
        self.debug_flags = debug_flags
        self.qp_solver = qp_solver  # "ipm" (cvxopt's coneqp iterate: the reference's behaviour, default) | "exact" (closed-form KKT point, opt-in)
        self.n_hdv = int(n_hdv)  # device reset: the last n_hdv vehicles of every env are IDM/MOBIL HDVs
        # draw_counts: every (re)spawn draws its vehicle counts as MergeEnv._num_vehicles does (config["traffic_density"]
        # 1..3, config["mixed_traffic"] / "traffic_type"); N is then the slot capacity of a ragged batch
        self.draw_counts, self.num_cav = bool(draw_counts), int(num_cav)
        self.n_f = 6 if env_id == "merge-multi-agent-v1" else 5
        self.n_s = 5 * self.n_f  # merge_env_v1.py:28 / :413
        self._cfg = self._make_cfg()
        self.T = int(self.config["duration"] * self.config["policy_frequency"])
        lay = self.layout = clib.state_layout(self.E, self.N)
        A = self.E * self.N
        dev = self.device
        raw = self._raw_state = torch.zeros(lay.total_bytes + 256, dtype=torch.uint8, device=dev)
        skew = (-raw.data_ptr()) % 256  # hipMalloc is 256-B aligned already; host tensors are not
        self.state = raw[skew: skew + lay.total_bytes]
        assert self.state.data_ptr() % 256 == 0
        self.f64 = self.state[lay.f64_offset: lay.f64_offset + 8 * A * len(abi.F_PLANES)].view(
            torch.float64).view(len(abi.F_PLANES), self.E, self.N)
        self.u8 = self.state[lay.u8_offset: lay.u8_offset + A * len(abi.B_PLANES)].view(
            len(abi.B_PLANES), self.E, self.N)
        self.env_i32 = self.state[lay.env_offset: lay.env_offset + 4 * self.E * len(abi.E_PLANES)].view(
            torch.int32).view(len(abi.E_PLANES), self.E)
        self.seeds = self.state[lay.seed_offset: lay.seed_offset + 8 * self.E].view(torch.int64)
        odt = torch.float64 if obs_f64 else torch.float32
        z = lambda *s, dtype=torch.float64: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        self.obs = z(self.E, self.N, self.n_s, dtype=odt)
        self.avail = z(self.E, self.N, 5, dtype=torch.uint8)
        self.out = {
            "reward": z(self.E), "done": z(self.E, dtype=torch.uint8),
            "agents_rewards": z(self.E, self.N), "regional_rewards": z(self.E, self.N),
            "agents_dones": z(self.E, self.N, dtype=torch.uint8), "agents_info": z(self.E, self.N, 3),
            "crashed": z(self.E, self.N, dtype=torch.uint8), "average_speed": z(self.E),
            "traffic_speed": z(self.E), "min_headway": z(self.E), "merge_percent": z(self.E),
            "action_mask": z(self.E, self.N, 5, dtype=torch.uint8),
        }
        # skip_outputs: per-agent planes a caller does not consume -- "agents_info" (24 B/agent), "action_mask" (5 B), "crashed"
        # (1 B): the kernel does not write an output whose MMStepOut pointer is NULL (mm_abi.h); step()'s info dict then has
        # no such key.  The reference's info always carries them, so the default (and MergeEnvCompat) keeps everything.
        for k in skip_outputs:
            if k not in ("agents_info", "action_mask", "crashed"):
                raise ValueError("only agents_info / action_mask / crashed may be skipped, got %r" % (k,))
            if k == "action_mask" and self.config.get("action_masking"):
                raise ValueError("action_masking is on: the mask is part of the step's result")
            del self.out[k]
        self.trace = z(3, len(abi.T_PLANES), self.E, self.N) if trace else None
        self.metrics = None
        self._step_out = abi.MMStepOut()  # (zero-initialised: a skipped output stays NULL)
        self._step_out.obs = self.obs.data_ptr()
        for k, t in self.out.items():
            setattr(self._step_out, k, t.data_ptr())
        self._step_out.trace = self.trace.data_ptr() if trace else None
        self._h = C.c_void_p()
        index = self.device.index if self.device.type == "cuda" else 0
        clib.check(clib.lib.mm_create(C.byref(self._cfg), self.E, self.N, index or 0,
                                      _ptr(self.state), lay.total_bytes, int(first_env),
                                      C.byref(self._h)))

    # -- configuration ------------------------------------------------------------------
    def _make_cfg(self):
        return abi.make_config(self.env_id, self.config, cbf_eta=self.cbf_eta, cbf_tau=self.cbf_tau,
                               auto_reset=self.auto_reset, obs_f64=self.obs_f64, seed=self.seed,
                               debug_flags=self.debug_flags, n_hdv=self.n_hdv, qp_solver=self.qp_solver,
                               draw_counts=self.draw_counts, num_cav=self.num_cav)

    def configure(self, config=None, **kw):
        """env.config[k] = v after construction (run_mappo.py:145-171); CBFType globals via kw."""
        if config:
            self.config.update(config)
        for k in ("cbf_eta", "cbf_tau", "auto_reset", "seed", "n_hdv", "qp_solver", "draw_counts", "num_cav"):
            if k in kw:
                setattr(self, k, kw[k])
        self._cfg = self._make_cfg()
        self.T = int(self.config["duration"] * self.config["policy_frequency"])
        self.clib.check(self.clib.lib.mm_set_config(self._h, C.byref(self._cfg)), self._h)

    def _stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return None

    # -- Env API --------------------------------------------------------------------------
    def reset(self, env_mask=None, seeds=None):
        """AbstractEnv.reset (abstract.py:176-209) with the device RNG, for all / masked envs."""
        if env_mask is not None:
            env_mask = env_mask.to(self.device, torch.uint8).contiguous()
        if seeds is not None:
            seeds = seeds.to(self.device, torch.int64).contiguous()
        self.clib.check(self.clib.lib.mm_reset(self._h, _ptr(env_mask), _ptr(seeds), _ptr(self.obs),
                                               _ptr(self.avail), self._stream()), self._h)
        return self.obs, self.avail

    def set_kinematics(self, x, y, heading, speed, n_merge=None, env_mask=None, kind=None):
        """Host-provided spawn (numpy-compatible reset, fixtures): [E,N] tensors; NaN x = absent.
        kind: optional [E,N] with 1 = controlled CAV, 2 = HDV (CAVs first)."""
        dev = self.device
        x = torch.as_tensor(x, dtype=torch.float64, device=dev).view(self.E, self.N)
        present = ~torch.isnan(x)
        sel = slice(None) if env_mask is None else env_mask.to(dev).bool()
        put = lambda plane, v: plane.__setitem__(sel, v[sel])  # noqa: E731
        put(self.f64[abi.F["X"]], torch.nan_to_num(x))
        put(self.f64[abi.F["Y"]], torch.as_tensor(y, dtype=torch.float64, device=dev).view(self.E, self.N))
        put(self.f64[abi.F["HEADING"]], torch.as_tensor(heading, dtype=torch.float64, device=dev).view(self.E, self.N))
        put(self.f64[abi.F["SPEED"]], torch.as_tensor(speed, dtype=torch.float64, device=dev).view(self.E, self.N))
        k8 = present.to(torch.uint8)
        if kind is not None:
            k8 = k8 * torch.as_tensor(kind, device=dev).view(self.E, self.N).to(torch.uint8)
            # the kernels that carry IDM / MOBIL run when HDVs can exist (the library's needs_general()): a fixed HDV count, a
            # count draw with mixed / "av" traffic, or steer_vel lateral control
            general = self.n_hdv > 0 or self._cfg.lateral_control != 0 or (self._cfg.traffic_density > 0 and self._cfg.mixed_traffic != 0)
            if not general and bool((k8 == 2).any()):
                # the CAV-only kernels would drive such a vehicle as a CAV from actions[i]
                raise ValueError("HDVs (kind == 2) in the spawn need mixed traffic: configure(n_hdv=...) > 0 first")
        put(self.u8[abi.B["KIND"]], k8)
        m = None if env_mask is None else env_mask.to(dev, torch.uint8).contiguous()
        self.clib.check(self.clib.lib.mm_init_from_kinematics(self._h, _ptr(m), self._stream()), self._h)
        if n_merge is not None:
            put(self.env_i32[abi.EP["N_MERGE"]], torch.as_tensor(n_merge, dtype=torch.int32, device=dev).view(self.E))
        return self.observe()

    def observe(self):
        self.clib.check(self.clib.lib.mm_observe(self._h, _ptr(self.obs), _ptr(self.avail),
                                                 self._stream()), self._h)
        return self.obs, self.avail

    def step(self, actions, obs_out=None, out=None):
        """MergeEnv.step (merge_env_v1.py:126-166) for every env; actions int32 [E, N] in 0..4.
        obs_out: optional caller buffer (same shape / dtype / device as self.obs, contiguous) the new
        observation is written to instead of self.obs -- a rollout hands over its states[t + 1] slot.
        out: optional {key: tensor} for keys of the info dict (same shape / dtype / device as self.out[key], contiguous):
        this step writes those outputs there instead of into self.out -- a rollout hands over rewards[t], dones[t], ...
        and saves a copy kernel per quantity and step; the returned info carries the given tensors under those keys."""
        abi.check_supervisor(self.config.get("safety_guarantee"))
        if actions.dtype != torch.int32 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(self.device, torch.int32).contiguous()
        assert actions.numel() == self.E * self.N
        obs, info = self.obs, self.out
        if obs_out is not None:
            assert obs_out.shape == self.obs.shape and obs_out.dtype == self.obs.dtype and obs_out.device == self.obs.device \
                and obs_out.is_contiguous()
            obs = obs_out
            self._step_out.obs = obs.data_ptr()
        if out:
            for k, t in out.items():  # (validate everything before any pointer of the call structure changes)
                ref = self.out[k]  # (KeyError: not an output of this env, or one it was built without)
                assert t.shape == ref.shape and t.dtype == ref.dtype and t.device == ref.device and t.is_contiguous(), k
            info = dict(self.out)
            for k, t in out.items():
                setattr(self._step_out, k, t.data_ptr())
                info[k] = t
        try:
            self.clib.check(self.clib.lib.mm_step(self._h, _ptr(actions), C.byref(self._step_out),
                                                  self._stream()), self._h)
        finally:
            self._step_out.obs = self.obs.data_ptr()
            for k in out or ():
                if k in self.out:
                    setattr(self._step_out, k, self.out[k].data_ptr())
        return obs, info["reward"], info["done"], info

    # -- checkpoint / resume --------------------------------------------------------------
    def state_dict(self):
        """Everything an env batch needs to resume bit-identically: the caller-owned state buffer (all planes,
        counters, per-env RNG seeds) and the last observation.  Config is the constructor's business."""
        d = {"abi_version": abi.MM_ABI_VERSION, "E": self.E, "N": self.N, "env_id": self.env_id,
             "state": self.state.clone(), "obs": self.obs.clone(), "avail": self.avail.clone()}
        if self.metrics is not None:
            d["metrics"] = self.flush_metrics().clone()  # (deferred metrics: what the partials hold belongs to the snapshot)
        return d

    def load_state_dict(self, d):
        if (d["abi_version"], d["E"], d["N"], d["env_id"]) != (abi.MM_ABI_VERSION, self.E, self.N, self.env_id):
            raise ValueError("checkpoint is for a different batch shape / env / ABI version")
        self.state.copy_(d["state"].to(self.device))
        self.obs.copy_(d["obs"].to(self.device))
        self.avail.copy_(d["avail"].to(self.device))
        if "metrics" in d and self.metrics is not None:
            self.flush_metrics()  # (pending deferred sums belong to the state that is being replaced)
            self.metrics.copy_(d["metrics"].to(self.device))

    def enable_metrics(self, deferred=False):
        """Device-side rollout metric accumulator: 7 sums + 1 min.
        deferred: the per-step sums stay in the library's per-wave partials and reach the returned tensor only in
        flush_metrics() / poll_errors() (mm_defer_metrics: one fold per rollout instead of one small launch per step)."""
        self.metrics = torch.zeros(8, dtype=torch.float64, device=self.device)
        self.metrics[7] = float("inf")
        self.clib.check(self.clib.lib.mm_set_metrics_buffer(self._h, _ptr(self.metrics)), self._h)
        self.clib.check(self.clib.lib.mm_defer_metrics(self._h, 1 if deferred else 0, self._stream()), self._h)
        self._metrics_deferred = bool(deferred)
        return self.metrics

    def flush_metrics(self):
        """Fold what deferred metrics hold into the tensor enable_metrics() returned (stream-ordered; no-op if not deferred)."""
        if self.metrics is not None:
            self.clib.check(self.clib.lib.mm_flush_metrics(self._h, self._stream()), self._h)
        return self.metrics

    def poll_errors(self):
        """Raise what the reference would have raised from inside step() since the last poll: check_bounds'
        ValueError (cbf.py:87-96) or an action outside 0..4.  Synchronises the stream (mm_poll_errors)."""
        self.clib.check(self.clib.lib.mm_poll_errors(self._h, self._stream()), self._h)

    def shield_qp(self, G, h, rows, solver=None, with_iters=False):
        """Batched stand-alone shield QP (cbf.py:110-161): G [n,4,3], h [n,4], rows [n] -> u [n,3], status
        (_cabi.QPS_*) [, IPM iterations].  solver: "exact" | "ipm" (default: this env's qp_solver)."""
        dev = self.device
        G = torch.as_tensor(G, dtype=torch.float64, device=dev).contiguous()
        h = torch.as_tensor(h, dtype=torch.float64, device=dev).contiguous()
        rows = torch.as_tensor(rows, dtype=torch.int32, device=dev).contiguous()
        n = rows.numel()
        u = torch.zeros(n, 3, dtype=torch.float64, device=dev)
        st = torch.zeros(n, dtype=torch.uint8, device=dev)
        it = torch.zeros(n, dtype=torch.int32, device=dev)
        sid = abi.qp_solver_id(self.qp_solver if solver is None else solver)
        self.clib.check(self.clib.lib.mm_shield_qp(self._h, n, _ptr(G), _ptr(h), _ptr(rows), sid, _ptr(u),
                                                   _ptr(st), _ptr(it), self._stream()), self._h)
        return (u, st, it) if with_iters else (u, st)

    def shield_actions(self, act_steer, act_acc):
        """safety_layer(...) (decentral_layer.py:767-817) for every controlled vehicle on the current state,
        each evaluated independently; nothing is stepped.  Inputs / outputs [E, N] float64.
        Returns (safe_steer, safe_acc, status uint8 bits _cabi.ST_*, lc_margin, headway) -- headway: what the call's
        vehicle.set_min_headway leaves in vehicle.min_headway (decentral_layer.py:466,700), NaN where the shield is gated off."""
        dev = self.device
        a_s = torch.as_tensor(act_steer, dtype=torch.float64, device=dev).reshape(self.E, self.N).contiguous()
        a_a = torch.as_tensor(act_acc, dtype=torch.float64, device=dev).reshape(self.E, self.N).contiguous()
        s_s, s_a = torch.empty_like(a_s), torch.empty_like(a_a)
        st = torch.zeros(self.E, self.N, dtype=torch.uint8, device=dev)
        mg, hw = torch.empty_like(a_s), torch.empty_like(a_s)
        self.clib.check(self.clib.lib.mm_shield_actions(self._h, _ptr(a_s), _ptr(a_a), _ptr(s_s), _ptr(s_a), _ptr(st),
                                                        _ptr(mg), _ptr(hw), self._stream()), self._h)
        return s_s, s_a, st, mg, hw

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.clib.lib.mm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reduce_rollout_metrics(metrics):
    """The path's only collective: SUM the 7 accumulators and MIN the headway over ranks.
    RCCL on GPUs (backend "nccl"), gloo on CPU; 64 bytes, latency-bound.  In place; returns metrics."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # gloo (CPU tests, 1-GPU rehearsals) reduces host tensors; RCCL reduces in place on the device
        via_host = metrics.is_cuda and dist.get_backend() == "gloo"
        sums, mn = metrics[:7].clone(), metrics[7:8].clone()
        if via_host:
            sums, mn = sums.cpu(), mn.cpu()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        metrics[:7] = sums.to(metrics.device)
        metrics[7:8] = mn.to(metrics.device)
    return metrics


def shard_range(E_total, rank, world):
    """Contiguous env shard of a rank: (first_env, count).  Envs are independent episodes, so the
    batch partitions with no data-path exchange; RNG streams are keyed by the global env index."""
    base, rem = divmod(int(E_total), int(world))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


_HIP = None


def hip_library():
    """Load libmm_hip.so once; fail loudly when it has not been built (no fallback)."""
    global _HIP
    if _HIP is None:
        _HIP = abi.CLib(HIP_LIB)
    return _HIP


class VecMergeEnv(BatchedMergeEnv):
    """The MI355X product path: HIP kernels over HBM-resident state."""

    def __init__(self, E, N, env_id="merge-multi-agent-v1", config=None, device="cuda:0", **kw):
        if not torch.cuda.is_available():
            raise RuntimeError("VecMergeEnv needs a ROCm GPU (libmm_hip.so has no CPU fallback)")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("VecMergeEnv runs on a GPU device, got %s" % device)
        torch.cuda.set_device(dev)
        super().__init__(hip_library(), E, N, env_id=env_id, config=config, device=dev, **kw)
