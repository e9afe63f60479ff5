"""CPU checks of the drop-in boundary: header <-> ctypes binding <-> exported symbols."""
import ctypes
import os
import re

import pytest

from marl_mass_amd import _cabi as abi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(REPO, "include", "mm_abi.h")).read()


def test_plane_tables_match_header():
    for prefix, table in (("MM_F_", abi.F_PLANES), ("MM_B_", abi.B_PLANES), ("MM_E_", abi.E_PLANES), ("MM_T_", abi.T_PLANES)):
        block = re.search(r"enum \{\s*%s[^}]*\}" % prefix, HEADER, re.S).group(0)
        block = re.sub(r"/\*.*?\*/", "", block, flags=re.S)
        names = [n for n in re.findall(r"%s([A-Z0-9_]+)" % prefix, block) if n != "COUNT"]
        dedup = []
        for n in names:
            if n not in dedup:
                dedup.append(n)
        assert dedup == table, prefix


def test_header_declares_every_bound_symbol():
    for sym in abi.CLib.SYMBOLS:
        assert re.search(r"\b%s\s*\(" % sym, HEADER), sym


@pytest.mark.parametrize("lib", ["marl-mass_amd/csrc/libmm_hip.so", "oracle/libmm_oracle.so"])
def test_library_exports_abi(lib):
    """Both implementations load and export every symbol of include/mm_abi.h (no compute calls)."""
    path = os.path.join(REPO, lib)
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    dll = ctypes.CDLL(path)
    for sym in abi.CLib.SYMBOLS:
        assert hasattr(dll, sym), (lib, sym)
    dll.mm_abi_version.restype = ctypes.c_int32
    assert dll.mm_abi_version() == abi.MM_ABI_VERSION
    lay = abi.MMStateLayout()
    dll.mm_state_layout.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(abi.MMStateLayout)]
    assert dll.mm_state_layout(4096, 8, ctypes.byref(lay)) == 0
    A = 4096 * 8
    assert lay.u8_offset >= A * 8 * len(abi.F_PLANES) and lay.total_bytes % 256 == 0


def test_struct_sizes():
    assert ctypes.sizeof(abi.MMConfig) == 10 * 4 + 9 * 8 + 8 + 8 * 4
    assert ctypes.sizeof(abi.MMStepOut) == 14 * 8


def test_shield_dispatch_strings():
    """safe_controller.py:229-241 / decentral_layer.py:767-817 string dispatch."""
    f = abi.shield_from_safety_guarantee
    assert f("none") == f(None) == f("priority") == f("dmc") == abi.SHIELD_NONE  # no VEHICLE-level shield
    for baseline in ("priority", "dmc"):  # abstract.py:460-464 action supervisors: out of scope, must not step silently
        with pytest.raises(NotImplementedError):
            abi.check_supervisor(baseline)
    abi.check_supervisor("none"), abi.check_supervisor("cbf-cav")
    assert f("cbf-av") == f("cbf-avs") == f("cbf-avs_cint") == f("cbf-hss") == abi.SHIELD_HSS
    assert f("cbf-cav") == f("cbf-mass") == abi.SHIELD_MASS
    with pytest.raises(ValueError):
        f("cbf-avlon")


def test_product_path_refuses_cpu():
    import torch
    from marl_mass_amd import VecMergeEnv
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        VecMergeEnv(4, 4)


def _create_rc(clib, E, N, device="cpu", **over):
    """mm_create's status for a given (E, N, config) with a correctly sized state buffer."""
    import torch
    cfg = abi.make_config("merge-multi-agent-v1", dict(abi.default_env_config("merge-multi-agent-v1"), safety_guarantee="none"))
    for k, v in over.items():
        setattr(cfg, k, v)
    lay = abi.MMStateLayout()
    if clib.lib.mm_state_layout(max(E, 1), min(max(N, 1), 16), ctypes.byref(lay)) != 0:
        return "layout"
    buf = torch.zeros(lay.total_bytes + 256, dtype=torch.uint8, device=device)
    ptr = buf.data_ptr() + (-buf.data_ptr()) % 256
    h = ctypes.c_void_p()
    rc = clib.lib.mm_create(ctypes.byref(cfg), E, N, 0, ctypes.c_void_p(ptr), lay.total_bytes, 0, ctypes.byref(h))
    if rc == 0:
        clib.lib.mm_destroy(h)
    return rc


def test_create_rejects_bad_sizes_and_configs():
    """Edge cases of the boundary (oracle build): empty batch, zero / too many vehicles, an HDV count that
    leaves no controlled vehicle, a foreign ABI version, an unknown shield id -> error codes, not crashes."""
    import oracle_env
    clib = oracle_env.library()
    assert _create_rc(clib, 4, 8) == 0
    assert _create_rc(clib, 1, 1) == 0 and _create_rc(clib, 3, 12) == 0        # smallest / largest env
    assert _create_rc(clib, 0, 8) != 0 and _create_rc(clib, -1, 8) != 0         # empty batch
    assert _create_rc(clib, 4, 0) != 0 and _create_rc(clib, 4, 13) != 0         # beyond the 6 + 6 spawn slots
    assert _create_rc(clib, 4, 8, n_hdv=8) != 0 and _create_rc(clib, 4, 8, n_hdv=-1) != 0
    assert _create_rc(clib, 4, 8, abi_version=abi.MM_ABI_VERSION + 1) != 0
    assert _create_rc(clib, 4, 8, shield=7) != 0
    assert _create_rc(clib, 4, 8, qp_solver=abi.QP_IPM) == 0 and _create_rc(clib, 4, 8, qp_solver=2) != 0


@pytest.mark.gpu
def test_create_rejects_bad_sizes_hip():
    from marl_mass_amd import hip_library
    clib = hip_library()
    assert _create_rc(clib, 4, 8, device="cuda") == 0
    for E, N, over in ((0, 8, {}), (4, 0, {}), (4, 13, {}), (4, 8, {"n_hdv": 8}), (4, 8, {"abi_version": 1}), (4, 8, {"shield": 9}),
                       (4, 8, {"qp_solver": 2})):
        assert _create_rc(clib, E, N, device="cuda", **over) != 0, (E, N, over)


def test_skipped_outputs_are_not_written_and_change_nothing_else():
    """MMStepOut: a NULL per-agent output is not written (mm_abi.h); VecMergeEnv(skip_outputs=...) hands NULL for
    agents_info / action_mask / crashed.  Everything else -- state, obs, the other outputs -- is what the full call gives."""
    import torch
    import oracle_env
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "cbf-cav", "HEADWAY_TIME": 0.5}, cbf_eta=0.03125, qp_solver="exact", cbf_tau=0.5,
              seed=5, auto_reset=True)
    full, lean = oracle_env.OracleEnv(32, 8, **kw), oracle_env.OracleEnv(32, 8, skip_outputs=("agents_info", "action_mask", "crashed"), **kw)
    full.reset(); lean.reset()
    g = torch.Generator().manual_seed(2)
    for t in range(12):
        a = torch.randint(0, 5, (32, 8), generator=g, dtype=torch.int32)
        of, rf, df, inf_ = full.step(a)
        ol, rl, dl, inl = lean.step(a)
        assert torch.equal(full.state, lean.state) and torch.equal(of, ol) and torch.equal(rf, rl) and torch.equal(df, dl)
        assert set(inf_) - set(inl) == {"agents_info", "action_mask", "crashed"}
        for k in inl:
            assert torch.equal(inf_[k].nan_to_num(), inl[k].nan_to_num()), k
    with pytest.raises(ValueError):
        oracle_env.OracleEnv(4, 4, skip_outputs=("reward",), **kw)
    with pytest.raises(ValueError):  # with masking on, the mask is a result of the step
        oracle_env.OracleEnv(4, 4, env_id="merge-multi-agent-v0", config={"action_masking": True}, skip_outputs=("action_mask",))


def test_deferred_metrics_calls_on_the_cpu_twin():
    """mm_defer_metrics / mm_flush_metrics (mm_abi.h) exist on both libraries; the CPU twin adds every step's sums to the
    caller's buffer directly, so deferral changes nothing there -- and without a metrics buffer the call is refused."""
    import torch
    import oracle_env
    kw = dict(env_id="merge-multi-agent-v1", config={"safety_guarantee": "none"}, seed=3, auto_reset=True)
    a, b = oracle_env.OracleEnv(16, 4, **kw), oracle_env.OracleEnv(16, 4, **kw)
    ma, mb = a.enable_metrics(), b.enable_metrics(deferred=True)
    a.reset(); b.reset()
    act = torch.full((16, 4), 3, dtype=torch.int32)
    for _ in range(8):
        a.step(act); b.step(act)
    assert torch.allclose(ma, b.flush_metrics(), rtol=1e-12, atol=0) and float(mb[4]) == 8 * 16  # (OpenMP sums: order varies)
    c = oracle_env.OracleEnv(4, 4, **kw)
    assert c.clib.lib.mm_defer_metrics(c._h, 1, None) == abi.MM_ERR_INVALID_ARG
    assert c.clib.lib.mm_defer_metrics(c._h, 0, None) == abi.MM_OK and c.clib.lib.mm_flush_metrics(c._h, None) == abi.MM_OK


def test_default_numerics_is_the_interior_point_iterate():
    """One default for every product entry point: the QP mode nobody names is cvxopt's interior-point iterate (the
    reference's behaviour, inside north_star's 1e-5); the closed-form KKT point is an explicit opt-in."""
    import oracle_env
    from marl_mass_amd import compat
    cfg = abi.default_env_config("merge-multi-agent-v1")
    assert abi.make_config("merge-multi-agent-v1", dict(cfg, safety_guarantee="cbf-cav")).qp_solver == abi.QP_IPM
    assert abi.make_config("merge-multi-agent-v1", dict(cfg, safety_guarantee="cbf-cav"), qp_solver="exact").qp_solver == abi.QP_EXACT
    env = oracle_env.OracleEnv(2, 4, config={"safety_guarantee": "cbf-cav"})
    assert env.qp_solver == "ipm" and env._cfg.qp_solver == abi.QP_IPM
    assert compat.CBFType.QP_SOLVER == "ipm"
