"""Device-side `_num_vehicles` (merge_env_v1.py:180-211, :476-495): every (re)spawn of an env draws its CAV / HDV counts,
so a batch is ragged and its composition changes from episode to episode -- the reference's training distribution
(traffic_density 1..3), which a fixed-N batch cannot produce."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_env
from golden_util import GOLDEN
from marl_mass_amd import _cabi as abi

RANGES = {1: ((1, 3), (1, 3)), 2: ((2, 4), (2, 4)), 3: ((4, 6), (3, 5))}  # inclusive (CAV, HDV), merge_env_v1.py:183-204


def _env(make, E, N, td, mixed, **kw):
    # mixed: False (traffic_type "cav") | True ("mixed") | "av" (one CAV among HDVs, merge_env_v1.py:485-489)
    cfg = {"safety_guarantee": kw.pop("shield", "none"), "HEADWAY_TIME": 0.5, "traffic_density": td,
           "traffic_type": mixed if isinstance(mixed, str) else ("mixed" if mixed else "cav"), "mixed_traffic": bool(mixed)}
    return make(E, N, env_id="merge-multi-agent-v1", config=cfg, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5, draw_counts=True, **kw)


@pytest.mark.parametrize("td,mixed", [(1, False), (1, True), (2, True), (3, False), (3, True)])
def test_count_distribution_matches_the_reference_draw(td, mixed):
    E, N = 9000, 11
    env = _env(oracle_env.OracleEnv, E, N, td, mixed, seed=17)
    env.reset()
    kind = env.u8[abi.B["KIND"]].numpy()
    n_cav, n_hdv = (kind == 1).sum(1), (kind == 2).sum(1)
    # vehicles are a prefix: CAVs first, then HDVs, then absent slots
    code = np.where(kind == 0, 3, kind)
    assert (np.diff(code, axis=1) >= 0).all()
    (c_lo, c_hi), (h_lo, h_hi) = RANGES[td]
    if mixed:
        assert n_cav.min() == c_lo and n_cav.max() == c_hi and n_hdv.min() == h_lo and n_hdv.max() == h_hi
        for arr, lo in ((n_cav, c_lo), (n_hdv, h_lo)):  # each uniform over its three values (np.random.choice)
            freq = np.bincount(arr - lo, minlength=3) / E
            assert np.abs(freq - 1 / 3).max() < 0.02, freq
        joint = np.bincount((n_cav - c_lo) * 3 + (n_hdv - h_lo), minlength=9) / E  # and independent
        assert np.abs(joint - 1 / 9).max() < 0.015
    else:  # CAV-only traffic: num_CAV + num_HDV controlled vehicles, no HDVs (:206-209)
        assert (n_hdv == 0).all() and n_cav.min() == c_lo + h_lo and n_cav.max() == c_hi + h_hi
        freq = np.bincount(n_cav - (c_lo + h_lo), minlength=5) / E
        assert np.abs(freq - np.array([1, 2, 3, 2, 1]) / 9).max() < 0.02, freq
    # the support is exactly what the reference's own reset draws (tests/golden/reset.json: drawn / mixed)
    ref = json.load(open(os.path.join(GOLDEN, "reset.json")))
    rows = [r for r in (ref["mixed"] if mixed else ref["drawn"]) if r["td"] == td and r.get("env", "merge-multi-agent-v1").endswith("v1")]
    for r in rows:
        assert (c_lo if mixed else c_lo + h_lo) <= r["n"] <= (c_hi if mixed else c_hi + h_hi)
    # n_merge = CAVs spawned on the ramp = n_cav - n_cav // 2 (a single CAV tosses a coin)
    nm = env.env_i32[abi.EP["N_MERGE"]].numpy()
    multi = n_cav != 1
    assert (nm[multi] == (n_cav - n_cav // 2)[multi]).all() and set(nm[~multi]) <= {0, 1}


def test_num_cav_override_and_capacity_check():
    env = _env(oracle_env.OracleEnv, 512, 6, 1, True, seed=3, num_cav=2)   # reset(num_CAV=2): only the HDV count is drawn
    env.reset()
    kind = env.u8[abi.B["KIND"]].numpy()
    assert ((kind == 1).sum(1) == 2).all() and set((kind == 2).sum(1)) == {1, 2, 3}
    with pytest.raises(ValueError):   # density 2 draws up to 8 vehicles: 6 slots cannot hold them
        _env(oracle_env.OracleEnv, 4, 6, 2, True).reset()
    with pytest.raises(ValueError):
        _env(oracle_env.OracleEnv, 4, 6, 4, True)


def test_capacity_is_checked_for_every_composition_the_configuration_can_draw():
    """The reference raises ValueError from np.random.choice(replace=False) when a road runs out of spawn points
    (merge_env_v1.py:284-320); here mm_create / mm_set_config refuse such a configuration up front -- incl. the
    reset(num_CAV=k) override and the per-road limit of six, which an auto-reset inside step() would otherwise meet."""
    with pytest.raises(ValueError, match="slots"):       # 8 CAVs + up to 5 HDVs in 12 slots
        _env(oracle_env.OracleEnv, 4, 12, 3, True, num_cav=8)
    with pytest.raises(ValueError, match="spawn points"):  # 7 + 5 fit the slots, but the ramp would need 4 + 3 = 7 points
        _env(oracle_env.OracleEnv, 4, 12, 3, True, num_cav=7)
    env = _env(oracle_env.OracleEnv, 64, 11, 3, True, num_cav=6, seed=9, auto_reset=True)  # the boundary that fits: 3 + 3 / 2 + 3
    env.reset()
    kind = env.u8[abi.B["KIND"]].numpy()
    assert ((kind == 1).sum(1) == 6).all() and (kind == 2).sum(1).max() == 5
    x = env.f64[abi.F["X"]].numpy()
    assert (x[kind != 0] > 0).all()  # every vehicle sits on a spawn point (a missing one would be x = noise around 0)
    env2 = oracle_env.OracleEnv(4, 12, config={"safety_guarantee": "none"})  # fixed counts, then re-configured to an impossible draw
    with pytest.raises(ValueError):
        env2.configure({"traffic_density": 3, "traffic_type": "mixed", "mixed_traffic": True}, draw_counts=True, num_cav=9)
    with pytest.raises(ValueError):   # fixed counts that the device spawn cannot place: 11 CAVs + 1 HDV need 7 ramp points
        oracle_env.OracleEnv(4, 12, config={"safety_guarantee": "none"}, n_hdv=1).reset()


def test_traffic_type_av_draws_one_cav_among_hdvs():
    E, N = 6000, 8
    env = _env(oracle_env.OracleEnv, E, N, 2, "av", seed=23)
    env.reset()
    kind = env.u8[abi.B["KIND"]].numpy()
    n_cav, n_hdv = (kind == 1).sum(1), (kind == 2).sum(1)
    assert (n_cav == 1).all() and n_hdv.min() == 3 and n_hdv.max() == 7   # (2..4) + (2..4) - 1
    freq = np.bincount(n_hdv - 3, minlength=5) / E
    assert np.abs(freq - np.array([1, 2, 3, 2, 1]) / 9).max() < 0.02, freq
    with pytest.raises(NotImplementedError):
        _env(oracle_env.OracleEnv, 4, 8, 2, "hdv")


def test_mixed_traffic_none_means_mixed_on_v0():
    """merge_env_v1.py:206-209 folds the HDVs into the CAVs only when mixed_traffic `is not None and not ...`;
    run_mappo.py's fallback for the key is None."""
    cfg = abi.make_config("merge-multi-agent-v0", dict(abi.default_env_config("merge-multi-agent-v0"), mixed_traffic=None, traffic_density=1),
                          draw_counts=True)
    assert cfg.mixed_traffic == 1


def test_ragged_auto_reset_changes_composition_between_episodes():
    E, N = 256, 8
    env = _env(oracle_env.OracleEnv, E, N, 2, True, seed=5, auto_reset=True, shield="cbf-cav")
    env.reset()
    k0 = env.u8[abi.B["KIND"]].clone()
    g = torch.Generator().manual_seed(1)
    p = torch.tensor([0.1, 0.6, 0.1, 0.1, 0.1])
    finished = torch.zeros(E, dtype=torch.bool)
    for t in range(105):
        a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
        obs, r, d, info = env.step(a)
        finished |= d.bool()
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
    assert finished.all()
    k1 = env.u8[abi.B["KIND"]]
    changed = ((k0 != k1).any(1)).float().mean()
    assert changed > 0.5, "most envs must have drawn a different composition for their second episode"
    assert ((k1 == 1).sum(1) >= 2).all() and ((k1 != 0).sum(1) <= 8).all()


@pytest.mark.gpu
@pytest.mark.parametrize("td,mixed,N,shield", [(1, False, 6, "cbf-cav"), (2, True, 8, "cbf-cav"), (3, False, 11, "cbf-avs_cint"),
                                               (3, True, 12, "cbf-cav"), (1, True, 7, "none"), (2, "av", 8, "cbf-cav")])
def test_ragged_auto_reset_soak_bit_exact_on_gpu(td, mixed, N, shield):
    """HIP == oracle, every bit, over three episodes of a ragged, re-drawn batch (counts, spawns, absent slots)."""
    from marl_mass_amd import VecMergeEnv
    oracle_env.set_math_mode(1)
    try:
        E = 1024
        kw = dict(seed=77, auto_reset=True, shield=shield, obs_f64=True)
        gpu = _env(lambda E_, N_, **k: VecMergeEnv(E_, N_, device="cuda:0", **k), E, N, td, mixed, **dict(kw))
        cpu = _env(oracle_env.OracleEnv, E, N, td, mixed, **dict(kw))
        og, _ = gpu.reset(); oc, _ = cpu.reset()
        assert torch.equal(gpu.u8.cpu(), cpu.u8) and torch.equal(og.cpu(), oc)
        g = torch.Generator().manual_seed(21)
        p = torch.tensor([0.15, 0.5, 0.15, 0.1, 0.1])
        for t in range(310):
            a = torch.multinomial(p, E * N, True, generator=g).view(E, N).int()
            og, rg, dg, ig = gpu.step(a.cuda())
            oc, rc, dc, ic = cpu.step(a)
            if t % 25 == 24 or t == 309:
                assert torch.equal(gpu.u8[abi.B["KIND"]].cpu(), cpu.u8[abi.B["KIND"]]), t
                present = cpu.u8[abi.B["KIND"]] != 0
                assert torch.equal(gpu.env_i32.cpu(), cpu.env_i32), t
                for plane in range(gpu.u8.shape[0]):
                    assert torch.equal(gpu.u8[plane].cpu()[present], cpu.u8[plane][present]), (t, plane)
                for plane in range(gpu.f64.shape[0]):
                    assert torch.equal(gpu.f64[plane].cpu()[present].nan_to_num(), cpu.f64[plane][present].nan_to_num()), (t, plane)
                assert torch.equal(og.cpu(), oc) and torch.equal(rg.cpu(), rc) and torch.equal(dg.cpu(), dc), t
                assert torch.equal(ig["regional_rewards"].cpu(), ic["regional_rewards"]), t
        assert int(gpu.env_i32[abi.EP["EPISODE"]].min()) >= 3
    finally:
        oracle_env.set_math_mode(0)
