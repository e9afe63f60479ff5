"""GPU parity tests: the HIP path (through the C ABI) vs the reference's golden tapes and vs the
CPU oracle on seeded random rollouts.  Flags bit-exact, floats within 1e-5 (north_star)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_env
from golden_util import GOLDEN, check_safety_layer_probes, episode_files, free_run, load_episode, replay, FLOAT_TOL
from marl_mass_amd import VecMergeEnv, _cabi as abi

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _portable_math():
    """The oracle evaluates include/mm_math.h (as the kernels do): HIP-vs-oracle must be bit-equal."""
    oracle_env.set_math_mode(1)
    yield
    oracle_env.set_math_mode(0)


def test_math_bits_cpu_vs_gpu():
    """Every elementary function + sqrt + division: identical bits on gfx950 and x86-64."""
    clib, olib = _gpu_env(1, 2).clib, oracle_env.library()
    g = torch.Generator().manual_seed(0)
    n = 1 << 20
    u = torch.rand(n, dtype=torch.float64, generator=g)
    ranges = {0: (-8, 8), 1: (-8, 8), 2: (-1.2, 1.2), 3: (-4, 4), 4: (-1, 1), 5: (-20, 0.5), 6: (1e-6, 100),
              7: (0, 1e6), 8: (-1e3, 1e3)}
    x2 = (torch.rand(n, dtype=torch.float64, generator=g) * 50 + 0.01)
    # fn 9: the constant-divisor division of the kernels vs true division, on the divisors in use
    consts = torch.tensor([300.0, 24.0, 90.0, 3.141592653589793, 2.5, 1000.0, 20.0, 1 / 15], dtype=torch.float64)
    ranges[9] = (-2e3, 2e3)
    for fn, (lo, hi) in ranges.items():
        x = (lo + (hi - lo) * u).contiguous()
        if fn == 9:
            x2 = consts[torch.arange(n) % len(consts)].contiguous()
        yc = torch.zeros(n, dtype=torch.float64)
        assert olib.lib.mm_math_eval(fn, n, x.data_ptr(), x2.data_ptr(), yc.data_ptr(), None) == 0
        xg, x2g, yg = x.cuda(), x2.cuda(), torch.zeros(n, dtype=torch.float64, device="cuda:0")
        assert clib.lib.mm_math_eval(fn, n, xg.data_ptr(), x2g.data_ptr(), yg.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(yg.cpu().view(torch.int64), yc.view(torch.int64)), "fn %d differs" % fn


def _gpu_env(E, N, **kw):
    return VecMergeEnv(E, N, device="cuda:0", **kw)


PARITY_JSON = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "golden_parity_gpu.json")


def _record_parity(name, err):
    """Per-tape error maxima of the GPU replay, kept where the driver / gpurun pull them from (gpurun_out/)."""
    os.makedirs(os.path.dirname(PARITY_JSON), exist_ok=True)
    try:
        with open(PARITY_JSON) as fh:
            rows = json.load(fh)
    except (OSError, ValueError):
        rows = {}
    rows[name] = {k: float(v) for k, v in err.items()}
    with open(PARITY_JSON, "w") as fh:
        json.dump(rows, fh, indent=1, sort_keys=True)


@pytest.mark.parametrize("path", episode_files(), ids=lambda p: os.path.basename(p)[:-4])
def test_golden_tape(path):
    """Every committed tape of the reference, teacher-forced, on the HIP path: floats within 1e-9, every discrete
    quantity exact -- NO knife-edge allowance on the committed tapes (the count is asserted to be 0)."""
    err = replay(_gpu_env, path, tol=1e-9, max_knife_edges=0)
    assert err["knife_edges"] == 0
    _record_parity(os.path.basename(path)[:-4], err)
    print(os.path.basename(path), json.dumps({k: float("%.3g" % v) for k, v in err.items()}))


@pytest.mark.parametrize("path", [p for p in episode_files() if os.path.basename(p).startswith("ipm_")], ids=lambda p: os.path.basename(p)[:-4])
def test_golden_tape_split_step(path):
    """The interior-point tapes once more through the SPLIT step (a one-env batch runs the fused kernel by itself: debug_flags
    bit3 forces the phase kernels + sweep kernel), same bar: the reference's tape directly, 1e-9, no knife-edges."""
    err = replay(lambda E, N, **kw: _gpu_env(E, N, debug_flags=8, **kw), path, tol=1e-9, max_knife_edges=0)
    assert err["knife_edges"] == 0


@pytest.mark.parametrize("path", episode_files("sc_*.npz"), ids=lambda p: os.path.basename(p)[:-4])
def test_crash_scenarios_free_running(path):
    """test/cbf crash scenarios on the GPU, free-running: crash step / no-crash as in the reference."""
    _, meta = load_episode(path)
    steps, crashed, mh = free_run(_gpu_env, path)
    assert (steps, crashed) == (meta["steps"], meta["crashed"])
    if meta["shield"] != "none":
        assert mh > 0.0


@pytest.mark.parametrize("path", episode_files("sl_*.npz"), ids=lambda p: os.path.basename(p)[:-4])
def test_standalone_safety_layer(path):
    """mm_shield_actions on the GPU vs the reference's per-vehicle safety_layer(...) calls."""
    worst, checked = check_safety_layer_probes(_gpu_env, path, tol=1e-9)
    assert checked > 0


def test_standalone_safety_layer_bits_vs_oracle():
    """Same entry, random states: HIP == oracle (mode 1) bit for bit, MASS with HDVs."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5},
              cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=31, auto_reset=True, n_hdv=3)
    E, N = 512, 8
    gpu, cpu = _gpu_env(E, N, **kw), oracle_env.OracleEnv(E, N, **kw)
    gpu.reset(); cpu.reset()
    g = torch.Generator().manual_seed(3)
    for t in range(30):
        a = torch.randint(0, 5, (E, N), generator=g, dtype=torch.int32)
        gpu.step(a.cuda()); cpu.step(a)
        if t % 5 == 4:
            steer = (torch.rand(E, N, dtype=torch.float64, generator=g) - 0.5) * 0.2
            acc = (torch.rand(E, N, dtype=torch.float64, generator=g) - 0.5) * 12
            rg, rc = gpu.shield_actions(steer, acc), cpu.shield_actions(steer, acc)
            for x, y in zip(rg, rc):
                assert torch.equal(x.cpu().nan_to_num(), y.nan_to_num()), t


CASES = [
    # (env_id, safety, N, E, steps, eta, tau)  -- BASELINE configs c2..c4 at oracle-friendly sizes
    ("merge-multi-agent-v0", "none", 4, 512, 40, 0.0, 1.2),
    ("merge-multi-agent-v1", "none", 4, 256, 40, 0.0, 1.2),
    ("merge-multi-agent-v1", "cbf-avs_cint", 4, 512, 100, 0.03125, 0.5),
    ("merge-multi-agent-v1", "cbf-cav", 8, 512, 100, 0.03125, 0.5),
    ("merge-multi-agent-v1", "cbf-cav", 5, 128, 60, 0.03125, 0.5),   # ragged group (N < G)
    ("merge-multi-agent-v1", "cbf-cav", 11, 128, 60, 0.03125, 0.5),  # G = 16
    ("merge-multi-agent-v1", "cbf-avs_cint", 2, 128, 60, 0.5, 1.2),
    ("merge-multi-agent-v1", "cbf-cav", 3, 128, 60, 0.03125, 0.5),        # ragged G = 4
    ("merge-multi-agent-v1", "cbf-avs_cint", 7, 128, 110, 0.03125, 0.5),  # ragged G = 8
    ("merge-multi-agent-v1", "cbf-cav", 12, 64, 110, 0.03125, 0.5),       # the 6 + 6 spawn-slot maximum
    ("merge-multi-agent-v1", "cbf-avs_cint", 9, 64, 60, 0.03125, 0.5),    # G = 16, HSS
    ("merge-multi-agent-v0", "none", 12, 64, 40, 0.0, 1.2),
    # mixed traffic: (N total vehicles, of which n_hdv IDM/MOBIL HDVs) -- 8th field
    ("merge-multi-agent-v0", "none", 6, 256, 60, 0.0, 1.2, 3),
    ("merge-multi-agent-v1", "none", 8, 256, 60, 0.0, 0.5, 4),
    ("merge-multi-agent-v1", "cbf-avs_cint", 7, 256, 110, 0.03125, 0.5, 3),
    ("merge-multi-agent-v1", "cbf-cav", 8, 512, 110, 0.03125, 0.5, 4),
    ("merge-multi-agent-v1", "cbf-cav", 11, 128, 110, 0.03125, 0.5, 5),
    # alternate agent rewards -- 9th field
    ("merge-multi-agent-v1", "cbf-cav", 8, 256, 110, 0.03125, 0.5, 0, "mrew"),
    ("merge-multi-agent-v1", "cbf-avs_cint", 6, 256, 110, 0.03125, 0.5, 2, "srew"),
    # lateral_control = "steer_vel" -- 10th field
    ("merge-multi-agent-v1", "cbf-cav", 8, 256, 110, 0.03125, 0.5, 0, "default", "steer_vel"),
    ("merge-multi-agent-v1", "cbf-avs_cint", 7, 256, 110, 0.03125, 0.5, 3, "default", "steer_vel"),
    ("merge-multi-agent-v1", "none", 4, 256, 60, 0.0, 0.5, 0, "default", "steer_vel"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(str(x) for x in c))
def test_random_rollout_vs_oracle(case):
    """Same seeds, same action tape, auto-reset on: every output of every step must agree."""
    env_id, safety, N, E, steps, eta, tau = case[:7]
    n_hdv = case[7] if len(case) > 7 else 0
    agent_reward = case[8] if len(case) > 8 else "default"
    lateral = case[9] if len(case) > 9 else "steer"
    kw = dict(env_id=env_id, config={"safety_guarantee": safety, "HEADWAY_TIME": tau, "agent_reward": agent_reward,
                                     "lateral_control": lateral},
              cbf_eta=eta, qp_solver="exact", cbf_tau=tau,
              obs_f64=True, seed=1000, auto_reset=True, n_hdv=n_hdv)
    gpu, cpu = _gpu_env(E, N, **kw), oracle_env.OracleEnv(E, N, **kw)
    og, ag = gpu.reset()
    oc, ac = cpu.reset()
    assert torch.equal(gpu.u8.cpu(), cpu.u8)
    assert torch.equal(gpu.f64.cpu()[:5], cpu.f64[:5]), "device reset must be bit-identical (integer RNG + exact fp)"
    assert float((og.cpu() - oc).abs().max()) <= 1e-12
    g = torch.Generator().manual_seed(123)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1])
    worst = 0.0
    for t in range(steps):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        og, rg, dg, ig = gpu.step(a.cuda())
        oc, rc, dc, ic = cpu.step(a)
        for k in ("done", "agents_dones", "crashed", "action_mask"):
            assert torch.equal(ig[k].cpu(), ic[k]), (t, k)
        assert torch.equal(gpu.u8.cpu(), cpu.u8), (t, "discrete state")
        assert torch.equal(gpu.env_i32.cpu(), cpu.env_i32), (t, "episode counters")
        errs = {"obs": (og.cpu() - oc).abs().max(), "state": (gpu.f64.cpu() - cpu.f64).nan_to_num().abs().max()}
        for k in ("reward", "agents_rewards", "regional_rewards", "average_speed", "traffic_speed", "min_headway"):
            errs[k] = (ig[k].cpu() - ic[k]).abs().max()
        mp_g, mp_c = ig["merge_percent"].cpu(), ic["merge_percent"]
        assert torch.equal(torch.isnan(mp_g), torch.isnan(mp_c))
        errs["merge"] = (mp_g - mp_c).nan_to_num().abs().max()
        m = max(float(v) for v in errs.values())
        worst = max(worst, m)
        assert m <= FLOAT_TOL, (t, {k: float(v) for k, v in errs.items()})
        # same elementary functions, same operation order, no contraction: every bit must agree
        assert m == 0.0, (t, {k: float(v) for k, v in errs.items()})
    print("bit-exact over %d steps x %d envs x %d agents" % (steps, E, N))


@pytest.mark.parametrize("safety,N", [("cbf-avs_cint", 8), ("cbf-avs_cint", 4), ("cbf-avs_cint", 11), ("cbf-avs", 6), ("cbf-cav", 8)])
def test_parallel_sweep_equals_literal_serial_sweep(safety, N):
    """The parallel fixed-point form of the shield sweep (what the CAV-only HSS and MASS kernels run) vs the literal
    front-to-back sweep (debug_flags bit0), LC-heavy action tape: identical bits everywhere."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": safety, "HEADWAY_TIME": 0.5},
              cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=4242, auto_reset=True, trace=True)
    E = 2048
    fast, slow = _gpu_env(E, N, **kw), _gpu_env(E, N, debug_flags=1, **kw)
    fast.reset()
    slow.reset()
    g = torch.Generator(device="cuda:0").manual_seed(11)
    p = torch.tensor([0.3, 0.2, 0.3, 0.1, 0.1], device="cuda:0")
    for t in range(110):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        fast.step(a)
        slow.step(a)
        assert torch.equal(fast.u8, slow.u8), t
        assert torch.equal(fast.f64.nan_to_num(), slow.f64.nan_to_num()), t
        assert torch.equal(fast.trace.nan_to_num(), slow.trace.nan_to_num()), t
        assert torch.equal(fast.obs, slow.obs) and torch.equal(fast.out["reward"], slow.out["reward"]), t


@pytest.mark.parametrize("qp,debug_flags,E", [("exact", 0, 1024), ("ipm", 4, 160), ("ipm", 8, 160)], ids=["exact", "ipm-fused", "ipm-split"])
@pytest.mark.parametrize("safety", ["cbf-cav", "cbf-avs_cint"])
def test_veto_passes_converge_from_any_first_guess(safety, qp, debug_flags, E):
    """The parallel form's veto passes (and the split step's slot selection) start from the veto each vehicle's shield decided
    one sub-step ago -- the `is_lc_safe` flag of the state -- as a FIRST GUESS.  The passes converge to the sequential answer
    from any guess: the same batch stepped from the same state with that flag bit scrambled gives identical bits everywhere
    (every present vehicle's flags are rewritten by the step's third sub-step at the latest; the default reward reads none)."""
    N = 8
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": safety, "HEADWAY_TIME": 0.5}, cbf_eta=0.03125,
              qp_solver=qp, cbf_tau=0.5, seed=777, auto_reset=True, trace=True, debug_flags=debug_flags)
    a, b = _gpu_env(E, N, **kw), _gpu_env(E, N, **kw)
    a.reset()
    b.reset()
    g = torch.Generator(device="cuda:0").manual_seed(5)
    p = torch.tensor([0.3, 0.2, 0.3, 0.1, 0.1], device="cuda:0")
    flipped = 0
    for t in range(60):
        act = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        b.state.copy_(a.state)
        flip = (torch.rand(b.u8[abi.B["KIND"]].shape, device="cuda:0", generator=g) < 0.5) & (b.u8[abi.B["KIND"]] != 0)
        b.u8[abi.B["FLAGS"]] ^= flip.to(torch.uint8) * abi.FLAG_IS_LC_SAFE
        flipped += int(flip.sum())
        a.step(act)
        b.step(act)
        assert torch.equal(a.u8, b.u8), t
        assert torch.equal(a.f64.nan_to_num(), b.f64.nan_to_num()), t
        assert torch.equal(a.env_i32, b.env_i32), t
        # (the trace's FLAGS plane shows the scrambled bit itself in the sub-steps before a new vehicle's first shield call)
        keep = [i for i, n in enumerate(abi.T_PLANES) if n != "FLAGS"]
        assert torch.equal(a.trace[:, keep].nan_to_num(), b.trace[:, keep].nan_to_num()), t
        assert torch.equal(a.obs, b.obs) and torch.equal(a.out["reward"], b.out["reward"]), t
    assert flipped > E * N * 10
    a.poll_errors()
    b.poll_errors()


@pytest.mark.parametrize("env_id,safety,N,E,n_hdv,lateral", [
    ("merge-multi-agent-v1", "cbf-cav", 6, 333, 0, "steer"), ("merge-multi-agent-v1", "cbf-cav", 11, 37, 0, "steer"),
    ("merge-multi-agent-v1", "cbf-cav", 12, 256, 0, "steer"), ("merge-multi-agent-v1", "cbf-avs_cint", 5, 1, 0, "steer"),
    ("merge-multi-agent-v1", "cbf-avs_cint", 9, 100, 0, "steer"), ("merge-multi-agent-v1", "none", 10, 64, 0, "steer"),
    ("merge-multi-agent-v0", "none", 6, 77, 0, "steer"), ("merge-multi-agent-v0", "none", 11, 10, 0, "steer"),
    # the general kernels: HDVs (IDM / MOBIL, the digital-twin slot of the literal sweep) and steer_vel
    ("merge-multi-agent-v1", "cbf-cav", 6, 333, 3, "steer"), ("merge-multi-agent-v1", "cbf-cav", 11, 77, 5, "steer"),
    ("merge-multi-agent-v1", "cbf-avs_cint", 12, 64, 6, "steer"), ("merge-multi-agent-v1", "none", 9, 50, 4, "steer"),
    ("merge-multi-agent-v0", "none", 5, 41, 2, "steer"), ("merge-multi-agent-v1", "cbf-cav", 10, 45, 0, "steer_vel"),
    ("merge-multi-agent-v1", "cbf-avs_cint", 6, 129, 2, "steer_vel")])
def test_six_and_twelve_lane_groups_equal_the_power_of_two_groups(env_id, safety, N, E, n_hdv, lateral):
    """Batches of 5..6 / 9..12 vehicles step in 6- / 12-lane groups (partner (a + m) mod G through ds_bpermute, ten /
    five envs per wave with four idle tail lanes) -- the same batch stepped in 8- / 16-lane groups (debug_flags bit1) must
    give identical bits everywhere: state, trace, observations, every output, the rollout metrics; batch sizes that leave
    the last wave partly empty, an env alone in its launch, action masking on (v0), mixed traffic, steer_vel."""
    kw = dict(env_id=env_id, config={"safety_guarantee": safety, "HEADWAY_TIME": 0.5, "lateral_control": lateral},
              cbf_eta=0.03125 if safety != "none" else 0.0, qp_solver="exact", cbf_tau=0.5, seed=99, auto_reset=True, trace=True, n_hdv=n_hdv)
    lanes, pow2 = _gpu_env(E, N, **kw), _gpu_env(E, N, debug_flags=2, **kw)
    ml, mp = lanes.enable_metrics(), pow2.enable_metrics(deferred=True)
    ol, al = lanes.reset(); op, ap = pow2.reset()
    assert torch.equal(ol, op) and torch.equal(al, ap)
    g = torch.Generator(device="cuda:0").manual_seed(5)
    p = torch.tensor([0.25, 0.25, 0.25, 0.15, 0.1], device="cuda:0")
    for t in range(130):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        ol, rl, dl, il = lanes.step(a)
        op, rp, dp, ip = pow2.step(a)
        assert torch.equal(lanes.u8, pow2.u8) and torch.equal(lanes.env_i32, pow2.env_i32), t
        assert torch.equal(lanes.f64.nan_to_num(nan=-7.0), pow2.f64.nan_to_num(nan=-7.0)), t
        assert torch.equal(lanes.trace.nan_to_num(nan=-7.0), pow2.trace.nan_to_num(nan=-7.0)), t
        assert torch.equal(ol, op) and torch.equal(rl, rp) and torch.equal(dl, dp), t
        for k in il:
            x, y = il[k], ip[k]
            assert torch.equal(x.nan_to_num(nan=-7.0) if x.is_floating_point() else x, y.nan_to_num(nan=-7.0) if y.is_floating_point() else y), (t, k)
    pow2.flush_metrics()
    torch.cuda.synchronize()
    assert torch.equal(ml[[1, 4, 6, 7]], mp[[1, 4, 6, 7]]) and torch.allclose(ml, mp, rtol=1e-12, atol=0)


def test_float32_obs_matches_float64():
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5},
              cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=7)
    e32, e64 = _gpu_env(256, 8, **kw), _gpu_env(256, 8, obs_f64=True, **kw)
    o32, _ = e32.reset()
    o64, _ = e64.reset()
    assert o32.dtype == torch.float32 and torch.equal(o32, o64.float())
    a = torch.ones(256, 8, dtype=torch.int32, device="cuda:0")
    for _ in range(5):
        o32 = e32.step(a)[0]
        o64 = e64.step(a)[0]
    assert torch.equal(o32, o64.float())


def test_shield_qp_entry_matches_golden(golden_dir):
    """mm_shield_qp on every (G, h) the reference assembled vs the exact-KKT u_bar in the tape."""
    z = np.load(os.path.join(golden_dir, "ep_v1_mass_N8_lc_s75.npz"))
    env = _gpu_env(1, 2, config={"safety_guarantee": "none"})
    u, st = env.shield_qp(z["qp_G"], np.nan_to_num(z["qp_h"]), z["qp_rows"], solver="exact")
    assert bool((st == abi.QPS_OPTIMAL).all())
    np.testing.assert_allclose(u.cpu().numpy()[:, 0], z["qp_x"][:, 0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(u.cpu().numpy()[:, 2], z["qp_x"][:, 2], rtol=0, atol=1e-9)


def test_shield_qp_ipm_bits_on_every_recorded_qp():
    """MM_QP_IPM on the GPU (one QP per lane, include/mm_qp.h) on EVERY (G, h) the reference assembled in any tape:
    d, s, status and iteration count equal, bit for bit, to what the reference-side coneqp stand-in returned
    (qp_x of the ipm_* tapes, qp_x_alt of the others) -- and to the oracle's general dense IPM."""
    from golden_util import is_ipm
    Gs, hs, rs, xs, sts, its = [], [], [], [], [], []
    for f in episode_files():
        z, meta = load_episode(f)
        if len(z["qp_rows"]) == 0:
            continue
        Gs.append(z["qp_G"]); hs.append(np.nan_to_num(z["qp_h"])); rs.append(z["qp_rows"])
        xs.append(z["qp_x"] if is_ipm(meta) else z["qp_x_alt"]); sts.append(z["qp_status"]); its.append(z["qp_iters"])
    G, h, rows = np.concatenate(Gs), np.concatenate(hs), np.concatenate(rs).astype(np.int32)
    x_ref, st_ref, it_ref = np.concatenate(xs), np.concatenate(sts), np.concatenate(its)
    assert len(rows) > 60000
    env = _gpu_env(1, 2, config={"safety_guarantee": "none"})
    u, st, it = env.shield_qp(G, h, rows, solver="ipm", with_iters=True)
    u, st, it = u.cpu().numpy(), st.cpu().numpy(), it.cpu().numpy()
    assert np.array_equal(u[:, 0].view(np.int64), x_ref[:, 0].view(np.int64)), "d differs from the reference-side IPM"
    assert np.array_equal(u[:, 2], x_ref[:, 2]) and np.array_equal(st, st_ref) and np.array_equal(it, it_ref)
    cpu = oracle_env.OracleEnv(1, 2, config={"safety_guarantee": "none"})
    uc, stc, itc = cpu.shield_qp(G, h, rows, solver="ipm", with_iters=True)
    assert np.array_equal(uc.numpy()[:, 0], u[:, 0]) and np.array_equal(stc.numpy(), st) and np.array_equal(itc.numpy(), it)
    print("IPM: %d QPs, %d 'unknown', iterations %d..%d" % (len(rows), int((st == 0).sum()), it.min(), it.max()))


def test_shield_qp_rejects_foreign_structure():
    """Only the G of get_G is supported (cbf.py:288-304,386-403): anything else is MM_ERR_INVALID_ARG, not a wrong answer."""
    env = _gpu_env(1, 2, config={"safety_guarantee": "none"})
    G = np.zeros((2, 4, 3)); G[:, 0] = (0.05, 0, -1); G[:, 1] = (1, 0, 0); G[:, 2] = (-1, 0, 0)
    h = np.array([[1.0, 2.0, 2.0, 0.0]] * 2)
    u, st = env.shield_qp(G, h, [3, 3], solver="exact")
    assert bool((st == abi.QPS_OPTIMAL).all())
    G[1, 0, 1] = 0.3  # a coupling with the steering variable the reference never builds
    with pytest.raises(ValueError):
        env.shield_qp(G, h, [3, 3], solver="exact")
    env.shield_qp(G[:1], h[:1], [3], solver="ipm")  # the latch was cleared by the failed call


# (last column: with the per-sub-step trace -- the trace-carrying and the production instantiation are different kernels)
IPM_CASES = [("cbf-cav", 8, 0, 256, 110, True), ("cbf-avs_cint", 4, 0, 256, 110, True), ("cbf-cav", 7, 3, 256, 110, True),
             ("cbf-avs_cint", 11, 0, 64, 60, True), ("cbf-cav", 2, 0, 128, 60, True),
             ("cbf-cav", 8, 0, 256, 110, False), ("cbf-avs_cint", 8, 0, 128, 60, False), ("cbf-cav", 4, 2, 128, 60, False),
             # CAV-only MASS / HSS in the 6- / 12-lane groups (N = 5..6, 9..12) and in 16-lane groups' vehicle counts
             ("cbf-cav", 6, 0, 130, 60, True), ("cbf-cav", 12, 0, 70, 60, True), ("cbf-cav", 5, 0, 64, 40, False),
             ("cbf-cav", 9, 0, 64, 40, False), ("cbf-avs_cint", 6, 0, 128, 40, False), ("cbf-avs_cint", 12, 0, 64, 40, True)]


@pytest.mark.parametrize("form", ["fused", "split"])
@pytest.mark.parametrize("safety,N,n_hdv,E,steps,trace", IPM_CASES, ids=lambda c: str(c))
def test_random_rollout_ipm_vs_oracle(safety, N, n_hdv, E, steps, trace, form):
    """Interior-point mode inside step(): the in-kernel interior-point QP (every vehicle, every sub-step) against the
    oracle's general dense IPM, free-running with auto-reset, LC-heavy tape: every bit of state / obs / rewards, and the
    per-sub-step status bits (is_optimal follows the IPM's status).  Both forms of the step: the fused kernel (what a batch
    this small runs by default) and the split step (phase kernels + lane-per-env sweep kernel)."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": safety, "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, cbf_tau=0.5,
              obs_f64=True, seed=2024, auto_reset=True, n_hdv=n_hdv, qp_solver="ipm", trace=trace)
    gpu, cpu = _gpu_env(E, N, debug_flags=8 if form == "split" else 4, **kw), oracle_env.OracleEnv(E, N, **kw)
    gpu.reset(); cpu.reset()
    g = torch.Generator().manual_seed(77)
    p = torch.tensor([0.25, 0.3, 0.25, 0.1, 0.1])
    for t in range(steps):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        og, rg, dg, ig = gpu.step(a.cuda())
        oc, rc, dc, ic = cpu.step(a)
        assert torch.equal(gpu.u8.cpu(), cpu.u8) and torch.equal(gpu.env_i32.cpu(), cpu.env_i32), t
        assert torch.equal(gpu.f64.cpu().nan_to_num(), cpu.f64.nan_to_num()), t
        assert torch.equal(og.cpu(), oc) and torch.equal(rg.cpu(), rc) and torch.equal(dg.cpu(), dc), t
        for plane in ("QP_ROWS", "QP_D", "STATUS", "SAFE_ACC") if trace else ():  # the IPM's iterate, its status and what was integrated
            k = abi.T[plane]
            assert torch.equal(gpu.trace[:, k].cpu().nan_to_num(nan=-7.0), cpu.trace[:, k].nan_to_num(nan=-7.0)), (t, plane)
    gpu.poll_errors(); cpu.poll_errors()  # check_bounds never fired


@pytest.mark.parametrize("safety,N,E,n_hdv", [("cbf-cav", 8, 200, 0), ("cbf-avs_cint", 4, 256, 0), ("cbf-cav", 11, 70, 0), ("cbf-cav", 6, 130, 0),
                                                ("cbf-avs_cint", 2, 64, 0), ("cbf-cav", 8, 128, 4), ("cbf-avs_cint", 7, 128, 3), ("cbf-cav", 11, 64, 5)])
def test_split_interior_point_step_equals_the_fused_kernel(safety, N, E, n_hdv):
    """The interior-point mode of a CAV-only batch steps as phase kernels + the lane-per-env sweep kernel.  Same bits -- state,
    per-sub-step trace (incl. every QP's rows and iterate), outputs -- as (a) the fused kernel with its wave-wide interior-point
    loop (debug_flags bit2) and (b) the sweep kernel classifying every ego itself instead of taking the phase kernel's slot
    selection (debug_flags bit0); LC-heavy action tape so that candidate-B commits (the case that invalidates the phase
    kernel's selection) occur.  Also (c) the sweep kernel with every speculation on a capped QP deliberately WRONG (debug_flags
    bit4: the assumed iterate is off by 2^-30, so every verification fails and the env is swept again literally -- the rollback
    path) and (d) with speculation switched off (bit5)."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": safety, "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, cbf_tau=0.5,
              qp_solver="ipm", seed=515, auto_reset=True, trace=True, obs_f64=True, n_hdv=n_hdv)
    # (a batch this small steps in the fused kernel by default: bit3 forces the split step)
    split, fused, slow = _gpu_env(E, N, debug_flags=8, **kw), _gpu_env(E, N, debug_flags=4, **kw), _gpu_env(E, N, debug_flags=8 | 1, **kw)
    wrong, nospec = _gpu_env(E, N, debug_flags=8 | 16, **kw), _gpu_env(E, N, debug_flags=8 | 32, **kw)
    for env in (split, fused, slow, wrong, nospec):
        env.reset()
    g = torch.Generator().manual_seed(3)
    p = torch.tensor([0.3, 0.2, 0.3, 0.1, 0.1])
    committed_b = capped = 0
    for t in range(70):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int().cuda()
        outs = [env.step(a) for env in (split, fused, slow, wrong, nospec)]
        capped += int((split.trace[:, abi.T["STATUS"]].nan_to_num().to(torch.int64) & (abi.ST_RAN | abi.ST_IS_OPTIMAL) == abi.ST_RAN).sum())
        for other, name in ((fused, "fused"), (slow, "self-classifying sweep"), (wrong, "wrong assumptions, re-swept"), (nospec, "no speculation")):
            assert torch.equal(split.u8, other.u8) and torch.equal(split.env_i32, other.env_i32), (t, name)
            assert torch.equal(split.f64.nan_to_num(), other.f64.nan_to_num()), (t, name)
            assert torch.equal(split.trace.nan_to_num(nan=-7.0), other.trace.nan_to_num(nan=-7.0)), (t, name)
        for o in outs[1:]:
            assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2]), t
            for k in outs[0][3]:
                assert torch.equal(outs[0][3][k].nan_to_num(), o[3][k].nan_to_num()), (t, k)
        # a veto that re-steers: safe steering differs from the nominal command
        tr = split.trace
        committed_b += int((tr[:, abi.T["SAFE_STEER"]].nan_to_num() != tr[:, abi.T["ACT_STEER"]].nan_to_num()).sum())
    assert committed_b > 0, "the tape must exercise candidate-B commits"
    if safety == "cbf-cav" and N <= 8 and not n_hdv:
        assert capped > 0, "the tape must contain QPs that run towards the iteration cap (the speculation's subject)"
    for env in (split, fused, slow, wrong, nospec):
        env.poll_errors()


def test_bad_action_is_latched():
    """An action outside 0..4 is a KeyError inside the reference's step (action.py:194-196): mm_step cannot return it
    from a launch, so it is latched and raised by poll_errors (the compat adapter polls after every step)."""
    env = _gpu_env(4, 4, config={"safety_guarantee": "none"})
    env.reset()
    a = torch.ones(4, 4, dtype=torch.int32, device="cuda:0")
    env.step(a); env.poll_errors()
    a[2, 1] = 7
    env.step(a)
    with pytest.raises(ValueError):
        env.poll_errors()
    env.step(torch.ones(4, 4, dtype=torch.int32, device="cuda:0")); env.poll_errors()  # cleared


def test_metrics_accumulator():
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "none"}, seed=3, auto_reset=True)
    env = _gpu_env(1024, 4, **kw)
    m = env.enable_metrics()
    env.reset()
    a = torch.full((1024, 4), 3, dtype=torch.int32, device="cuda:0")
    tot_r, tot_done, mn = 0.0, 0, float("inf")
    for _ in range(30):
        _, r, d, info = env.step(a)
        tot_r += float(r.sum())
        tot_done += int(d.sum())
        mn = min(mn, float(info["min_headway"].min()))
    mm = m.cpu()
    assert abs(float(mm[0]) - tot_r) <= 1e-6 * max(1.0, abs(tot_r))
    assert int(mm[6]) == tot_done and int(mm[4]) == 30 * 1024
    assert float(mm[7]) == mn


def test_deferred_metrics_match_per_step_folding():
    """mm_defer_metrics: the per-wave partials accumulate across steps and reach the caller's 8 doubles only in
    mm_flush_metrics / mm_poll_errors -- same totals as the per-step fold (sums up to reassociation, min and counts exactly),
    nothing visible before the flush, nothing counted twice by a second flush, and switching deferral off flushes."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact",
              cbf_tau=0.5, seed=5, auto_reset=True)
    a_env, b_env = _gpu_env(777, 8, **kw), _gpu_env(777, 8, **kw)   # (777 envs: the last wave of a launch is partly empty)
    ma, mb = a_env.enable_metrics(), b_env.enable_metrics(deferred=True)
    a_env.reset(); b_env.reset()
    g = torch.Generator().manual_seed(9)
    p = torch.tensor([0.2, 0.3, 0.2, 0.2, 0.1])
    for t in range(120):   # (episodes end and re-spawn inside the window: crashed / merge % / episode counters move)
        act = torch.multinomial(p, 777 * 8, True, generator=g).view(777, 8).int().cuda()
        a_env.step(act); b_env.step(act)
        if t == 50:
            torch.cuda.synchronize()
            assert float(mb[4]) == 0.0 and float(mb[7]) == float("inf"), "deferred sums became visible before a flush"
            b_env.flush_metrics(); b_env.flush_metrics()   # (the second flush has nothing left to add)
            torch.cuda.synchronize()
            assert float(mb[4]) == float(ma[4]) == 51 * 777
    b_env.poll_errors()   # flushes, then synchronises
    assert torch.equal(ma[[1, 4, 6, 7]], mb[[1, 4, 6, 7]]) and float(ma[6]) > 0
    assert torch.allclose(ma, mb, rtol=1e-12, atol=0)
    # a snapshot taken in deferred mode contains the pending sums; turning deferral off flushes and goes back to per-step folds
    b_env.step(act); a_env.step(act)
    assert torch.allclose(b_env.state_dict()["metrics"], ma, rtol=1e-12, atol=0)
    b_env.step(act); a_env.step(act)
    b_env.clib.check(b_env.clib.lib.mm_defer_metrics(b_env._h, 0, b_env._stream()), b_env._h)
    b_env.step(act); a_env.step(act)
    torch.cuda.synchronize()
    assert torch.allclose(ma, mb, rtol=1e-12, atol=0) and float(mb[4]) == 123 * 777


@pytest.mark.parametrize("E,N,density,qp,steps", [(65536, 8, 0, "exact", 12), (32768, 12, 0, "exact", 12), (65536, 6, 1, "exact", 12),
                                                  (32768, 11, 3, "exact", 12), (65536, 8, 0, "ipm", 12), (16384, 11, 3, "ipm", 12)],
                         ids=["headline", "12-lane-groups", "density1-6-lane", "density3-12-lane", "headline-ipm", "density3-12-lane-ipm"])
def test_full_size_bit_exact_vs_oracle(E, N, density, qp, steps):
    """BASELINE's headline size (65536 envs x 8 CAVs, MASS, auto-reset) for 12 steps against the
    OpenMP oracle: every state bit, obs, reward, done of all 524 288 agents -- and the bench sizes of the 6- / 12-lane
    group layouts: 32 768 x 12, and the reference's traffic_density 1 / 3 with the vehicle counts drawn per episode."""
    cfg = {"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}
    kw = dict(env_id="merge-multi-agent-v1", config=cfg, cbf_eta=0.03125, qp_solver=qp, cbf_tau=0.5, seed=1000, auto_reset=True)
    if density:
        cfg.update({"traffic_density": density, "traffic_type": "cav", "mixed_traffic": False})
        kw["draw_counts"] = True
    oracle_env.library().lib.orc_set_threads(16)
    gpu, cpu = _gpu_env(E, N, **kw), oracle_env.OracleEnv(E, N, **kw)
    gpu.reset()
    cpu.reset()
    # start mid-episode so the batch holds every phase of an episode (steps 0..99)
    ph = (torch.arange(E, dtype=torch.int32) * 37) % 90
    gpu.env_i32[abi.EP["STEPS"]] = ph.cuda()
    cpu.env_i32[abi.EP["STEPS"]] = ph
    g = torch.Generator().manual_seed(9)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1])
    for t in range(steps):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        og, rg, dg, ig = gpu.step(a.cuda())
        oc, rc, dc, ic = cpu.step(a)
        assert torch.equal(gpu.u8.cpu(), cpu.u8) and torch.equal(gpu.env_i32.cpu(), cpu.env_i32), t
        assert torch.equal(gpu.f64.cpu().nan_to_num(), cpu.f64.nan_to_num()), t
        assert torch.equal(og.cpu(), oc) and torch.equal(rg.cpu(), rc) and torch.equal(dg.cpu(), dc), t
        for k in ("agents_rewards", "regional_rewards", "min_headway", "average_speed", "crashed"):
            assert torch.equal(ig[k].cpu(), ic[k]), (t, k)
    assert int(gpu.env_i32[abi.EP["EPISODE"]].max()) >= 2, "some envs must have auto-reset"
    gpu.poll_errors(); cpu.poll_errors()


@pytest.mark.parametrize("name,env_id,safety,E,N,eta,tau,qp,steps", [
    ("c2", "merge-multi-agent-v0", "none", 4096, 4, 0.0, 1.2, "exact", 30),
    ("c3", "merge-multi-agent-v1", "cbf-avs_cint", 4096, 4, 0.03125, 0.5, "exact", 30),
    ("c4", "merge-multi-agent-v1", "cbf-cav", 16384, 8, 0.03125, 0.5, "exact", 30),
    ("c5-per-gpu", "merge-multi-agent-v1", "cbf-cav", 8192, 8, 0.03125, 0.5, "exact", 30),
    # the same configurations on the default numerics (interior-point QP: the split step of the CAV-only batches)
    ("c3-ipm", "merge-multi-agent-v1", "cbf-avs_cint", 4096, 4, 0.03125, 0.5, "ipm", 15),
    ("c4-ipm", "merge-multi-agent-v1", "cbf-cav", 16384, 8, 0.03125, 0.5, "ipm", 12),
    ("c5-per-gpu-ipm", "merge-multi-agent-v1", "cbf-cav", 8192, 8, 0.03125, 0.5, "ipm", 15)])
def test_baseline_configs_bit_exact(name, env_id, safety, E, N, eta, tau, qp, steps):
    """BASELINE.json configs c2..c5 at their own sizes: 30 steps (interior-point mode: 12 - 15) with the bench's
    action distribution, every output of every step equal to the oracle's."""
    kw = dict(env_id=env_id, config={"safety_guarantee": safety, "HEADWAY_TIME": tau}, cbf_eta=eta, qp_solver=qp, cbf_tau=tau,
              seed=1000, auto_reset=True)
    oracle_env.library().lib.orc_set_threads(16)
    gpu, cpu = _gpu_env(E, N, **kw), oracle_env.OracleEnv(E, N, **kw)
    gpu.reset()
    cpu.reset()
    ph = (torch.arange(E, dtype=torch.int32) * 13) % 95  # every phase of an episode is present
    gpu.env_i32[abi.EP["STEPS"]] = ph.cuda()
    cpu.env_i32[abi.EP["STEPS"]] = ph
    g = torch.Generator().manual_seed(17)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1])
    for t in range(steps):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        og, rg, dg, ig = gpu.step(a.cuda())
        oc, rc, dc, ic = cpu.step(a)
        assert torch.equal(og.cpu(), oc) and torch.equal(rg.cpu(), rc) and torch.equal(dg.cpu(), dc), (name, t)
        for k in ("agents_rewards", "regional_rewards", "agents_dones", "average_speed", "traffic_speed", "min_headway", "crashed", "action_mask"):
            assert torch.equal(ig[k].cpu(), ic[k]), (name, t, k)
    assert torch.equal(gpu.u8.cpu(), cpu.u8) and torch.equal(gpu.f64.cpu().nan_to_num(), cpu.f64.nan_to_num())
    gpu.poll_errors(); cpu.poll_errors()


@pytest.mark.parametrize("safety,n_hdv,lateral,N", [("cbf-cav", 0, "steer", 8), ("cbf-avs_cint", 0, "steer", 8), ("cbf-cav", 3, "steer", 8),
                                                      ("cbf-cav", 0, "steer_vel", 8), ("cbf-cav", 0, "steer", 4), ("cbf-cav", 0, "steer", 12),
                                                      ("cbf-avs_cint", 0, "steer", 2), ("cbf-cav", 5, "steer", 11), ("none", 2, "steer", 4),
                                                      # the parallel-form kernels (HSS, CAV-only, N <= 8) at every ragged size
                                                      ("cbf-avs_cint", 0, "steer", 3), ("cbf-avs_cint", 0, "steer", 4), ("cbf-avs_cint", 0, "steer", 5),
                                                      ("cbf-avs_cint", 0, "steer", 6), ("cbf-avs_cint", 0, "steer", 7), ("cbf-avs", 0, "steer", 8)])
def test_soak_three_episodes_bit_exact(safety, n_hdv, lateral, N):
    """Long free-running soak: 4096 envs x N vehicles x 320 steps (three full episodes with auto-reset),
    checked against the oracle every 40 steps and at the end -- ~10 M agent-steps per case, every bit,
    for every group size (G = 2, 4, 8, 16) and both kernel families (CAV-only / general)."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": safety, "HEADWAY_TIME": 0.5, "lateral_control": lateral},
              cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=77, auto_reset=True, n_hdv=n_hdv)
    E = 4096
    oracle_env.library().lib.orc_set_threads(16)
    gpu, cpu = _gpu_env(E, N, **kw), oracle_env.OracleEnv(E, N, **kw)
    gpu.reset()
    cpu.reset()
    g = torch.Generator().manual_seed(21)
    p = torch.tensor([0.15, 0.5, 0.15, 0.1, 0.1])
    for t in range(320):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        og, rg, dg, ig = gpu.step(a.cuda())
        oc, rc, dc, ic = cpu.step(a)
        if t % 40 == 39 or t == 319:
            assert torch.equal(gpu.u8.cpu(), cpu.u8) and torch.equal(gpu.env_i32.cpu(), cpu.env_i32), t
            assert torch.equal(gpu.f64.cpu().nan_to_num(), cpu.f64.nan_to_num()), t
            assert torch.equal(og.cpu(), oc) and torch.equal(rg.cpu(), rc) and torch.equal(dg.cpu(), dc), t
            assert torch.equal(ig["regional_rewards"].cpu(), ic["regional_rewards"]), t
    assert int(gpu.env_i32[abi.EP["EPISODE"]].min()) >= 3


def test_full_size_properties():
    """BASELINE c4/c5-sized batch (65536 envs x 8, MASS): size-independent invariants.
    With the shield on and eta=0.03125, tau=0.5 the reference never crashes under the random tape
    (SURVEY App. B), and sharding the batch must not change any env."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5},
              cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, seed=1000)
    E, N = 65536, 8
    env = _gpu_env(E, N, **kw)
    half = VecMergeEnv(E // 2, N, device="cuda:0", first_env=E // 2, **kw)
    env.reset()
    half.reset()
    assert torch.equal(env.f64[:, E // 2:].nan_to_num(), half.f64.nan_to_num()), "RNG streams are keyed by global env id"
    g = torch.Generator(device="cuda:0").manual_seed(5)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1], device="cuda:0")
    for t in range(25):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        obs, r, d, info = env.step(a)
        half.step(a[E // 2:].contiguous())
        assert not bool(info["crashed"].any()), "MASS-shielded env crashed at step %d" % t
        assert bool(torch.isfinite(obs).all()) and bool(torch.isfinite(r).all())
    assert torch.equal(env.f64[:, E // 2:].nan_to_num(), half.f64.nan_to_num()) and torch.equal(env.u8[:, E // 2:], half.u8)
    xs = env.f64[abi.F["X"]]
    assert bool((xs > 0).all()) and bool((env.f64[abi.F["SPEED"]] >= 0).all())


@pytest.mark.parametrize("safety,with_hdv", [("cbf-cav", False), ("cbf-avs_cint", False), ("cbf-cav", True), ("none", True)])
def test_ragged_batch_with_absent_slots(safety, with_hdv):
    """Envs of ONE batch with different vehicle counts (absent slots = NaN x, as the compat adapter pads a
    12-slot env): 2..8 vehicles per env, optionally a CAV prefix followed by HDVs, host-provided spawn on
    both sides, 40 steps, every bit equal to the oracle's."""
    E, N = 512, 8
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": safety, "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact",
              cbf_tau=0.5, seed=5, n_hdv=3 if with_hdv else 0)
    gpu, cpu = _gpu_env(E, N, **kw), oracle_env.OracleEnv(E, N, **kw)
    cpu.reset()  # a valid spawn to carve the ragged batch from
    gpu.reset()  # (same counters on both sides)
    x, y = cpu.f64[abi.F["X"]].clone(), cpu.f64[abi.F["Y"]].clone()
    h, v = cpu.f64[abi.F["HEADING"]].clone(), cpu.f64[abi.F["SPEED"]].clone()
    g = torch.Generator().manual_seed(8)
    count = torch.randint(2, N + 1, (E,), generator=g)
    slot = torch.arange(N)[None, :]
    x[slot >= count[:, None]] = float("nan")
    kind = torch.ones(E, N, dtype=torch.int64)
    if with_hdv:  # CAVs first, then HDVs: at least one CAV
        n_cav = torch.maximum(torch.ones_like(count), count - torch.randint(0, 4, (E,), generator=g))
        kind[slot >= n_cav[:, None]] = 2
    og, _ = gpu.set_kinematics(x.cuda(), y.cuda(), h.cuda(), v.cuda(), kind=kind.cuda())
    oc, _ = cpu.set_kinematics(x, y, h, v, kind=kind)
    assert torch.equal(og.cpu(), oc) and torch.equal(gpu.u8.cpu(), cpu.u8) and torch.equal(gpu.env_i32.cpu(), cpu.env_i32)
    p = torch.tensor([0.15, 0.5, 0.15, 0.1, 0.1])
    for t in range(40):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        og, rg, dg, ig = gpu.step(a.cuda())
        oc, rc, dc, ic = cpu.step(a)
        assert torch.equal(og.cpu(), oc) and torch.equal(rg.cpu(), rc) and torch.equal(dg.cpu(), dc), t
        assert torch.equal(ig["regional_rewards"].cpu(), ic["regional_rewards"]) and torch.equal(ig["min_headway"].cpu(), ic["min_headway"]), t
    assert torch.equal(gpu.u8.cpu(), cpu.u8) and torch.equal(gpu.f64.cpu().nan_to_num(), cpu.f64.nan_to_num())


@pytest.mark.parametrize("name", ["ep_v1_mass_N8_s0", "mx_v1_hss_4c3h_s25", "sv_v1_mass_N8_s50", "ipm_v1_mass_N8_s0", "ipm_v1_mass_4c3h_s25"])
def test_compat_adapter_on_gpu_matches_golden(name):
    """The drop-in object API (MergeEnvCompat: numpy-RNG reset replay + step tuple + control profile) over the
    HIP backend, free-running on a reference tape: what a maintainer's `env = make(env_id)` swap executes."""
    from marl_mass_amd import compat
    from golden_util import is_ipm
    z, meta = load_episode(os.path.join(GOLDEN, name + ".npz"))
    saved = (compat.CBFType.GAMMA_B, compat.CBFType.TAU, compat.CBFType.QP_SOLVER)
    compat.CBFType.GAMMA_B, compat.CBFType.TAU = meta["eta"], meta["headway_time"]
    # what answered solvers.qp while the reference produced the tape (the adapter's default is "ipm": the ipm_* tapes run it)
    compat.CBFType.QP_SOLVER = "ipm" if is_ipm(meta) else "exact"
    try:
        _compat_replay(compat, z, meta)
    finally:
        compat.CBFType.GAMMA_B, compat.CBFType.TAU, compat.CBFType.QP_SOLVER = saved


def _compat_replay(compat, z, meta):
    env = compat.make(meta["env_id"], store_profile=True)
    env.config.update({"safety_guarantee": meta["shield"], "HEADWAY_TIME": meta["headway_time"], "action_masking": False,
                       "lateral_control": meta.get("lateral_control", "steer")})
    env._num_vehicles = lambda num_CAV=0: (meta["n"], meta.get("n_hdv", 0))
    obs, avail = env.reset(is_training=False, testing_seeds=meta["seed"])
    np.testing.assert_allclose(obs, z["obs0"], rtol=0, atol=1e-12)
    for t in range(meta["steps"]):
        obs, reward, done, info = env.step(tuple(int(a) for a in z["actions"][t]))
        np.testing.assert_allclose(obs, z["obs"][t], rtol=0, atol=1e-9)
        assert abs(reward - z["reward"][t]) <= 1e-9 and done == bool(z["done"][t])
        np.testing.assert_allclose(info["regional_rewards"], z["regional_rewards"][t], rtol=0, atol=1e-9)
    assert done and env.is_crashed() == meta["crashed"]
    cp = env.control_profile()
    assert len(cp["av0"]["state_hist"]) == int(z["sub_count"].sum())
    assert abs(cp["av0"]["state_hist"][-1]["x"] - z["sub_f"][-1][0][0]) <= 1e-9


def test_skipped_outputs_on_gpu():
    """VecMergeEnv(skip_outputs=...): NULL MMStepOut pointers are not written by the step kernel, everything else is
    bit-identical to the full call (what bench.py's headline line requests)."""
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5,
              seed=5, auto_reset=True)
    full, lean = _gpu_env(512, 8, **kw), _gpu_env(512, 8, skip_outputs=("agents_info", "action_mask", "crashed"), **kw)
    full.reset(); lean.reset()
    g = torch.Generator().manual_seed(2)
    for t in range(30):
        a = torch.randint(0, 5, (512, 8), generator=g, dtype=torch.int32).cuda()
        of, rf, df, inf_ = full.step(a)
        ol, rl, dl, inl = lean.step(a)
        assert torch.equal(full.state, lean.state) and torch.equal(of, ol) and torch.equal(rf, rl) and torch.equal(df, dl)
        assert set(inf_) - set(inl) == {"agents_info", "action_mask", "crashed"}
        for k in inl:
            assert torch.equal(inf_[k].nan_to_num(), inl[k].nan_to_num()), k


def test_geom_unit_tables_on_the_device():
    """The reference's unit tables (tests/golden/units.npz) against the DEVICE functions of the step kernel -- closest_lane
    (with its kb0 pruning), next_lane, reachability, steering_control, speed_to_index, and the collision decision THROUGH
    the kernel's own exact early-out boxes_may_touch -- not only through trajectories (mm_geom_eval, include/mm_abi.h)."""
    import test_oracle_golden as tog
    from marl_mass_amd import vec_env
    units = np.load(os.path.join(GOLDEN, "units.npz"))
    clib = abi.CLib(vec_env.HIP_LIB)
    tog.check_geom_tables(clib, units, device="cuda:0")
    # randomised near-contact sweep: the early-out never suppresses a hit the 9-point test reports, for vehicles and for
    # the obstacle; and the device's 9-point test agrees with the oracle's (libm) on every pair that is not a knife-edge
    rows = tog.near_contact_rows()
    dev = tog._geom(clib, abi.GEOM_RECT, rows, 4, "cuda:0")
    d2 = (rows[:, 3] - rows[:, 0]) ** 2 + (rows[:, 4] - rows[:, 1]) ** 2
    pre = ~(np.sqrt(d2) > 5.0)
    assert np.array_equal(dev[:, 0], pre & (dev[:, 2] != 0)) and np.array_equal(dev[:, 1], pre & (dev[:, 3] != 0))
    assert dev[:, 2].sum() > 20000 and (pre & (dev[:, 2] == 0)).sum() > 20000  # both outcomes are well represented
    oracle_env.set_math_mode(1)
    try:
        cpu = tog._geom(oracle_env.library(), abi.GEOM_RECT, rows, 4)
    finally:
        oracle_env.set_math_mode(0)
    assert np.array_equal(dev, cpu)  # same arithmetic (mm_math) on both sides: every flag equal
