"""The frozen-iterate certificate of include/mm_qp.h (mm_qp_frozen): the device builds stop a QP that would run to cvxopt's
100-iteration cap as soon as its iterate provably cannot change any more.  Here, on the CPU: the certified loop returns the
same (d, slack, status, iteration count) bits as the literal loop on every QP the reference assembled in any tape and on
random QPs around the feasibility boundary of the CBF row (tools/qp_certificate_fuzz.c).  The GPU parity tests compare the
HIP kernels (certified) with the oracle (literal loop) on top of this."""
import json
import os
import subprocess

import numpy as np
import pytest

from golden_util import episode_files, load_episode

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("qpcert") / "qp_certificate_fuzz")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-o", out, os.path.join(REPO, "tools", "qp_certificate_fuzz.c"), "-lm"])
    return out


def test_certificate_on_every_recorded_qp(fuzz_bin):
    lines = []
    for f in episode_files():
        z, _ = load_episode(f)
        if len(z["qp_rows"]) == 0:
            continue
        G, h, rows = z["qp_G"], np.nan_to_num(z["qp_h"]), z["qp_rows"]
        lines += ["%r %r %r %r %r %d" % (float(G[i, 0, 0]), float(h[i, 0]), float(h[i, 1]), float(h[i, 2]), float(h[i, 3]), int(rows[i]))
                  for i in range(len(rows))]
    assert len(lines) > 60000
    res = json.loads(subprocess.run([fuzz_bin, "-"], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout)
    assert res["qps"] == len(lines) and res["mismatches"] == 0
    # every QP of the tapes that runs to the cap is recognised, and none before iteration 20
    assert res["capped"] > 300 and res["capped_not_certified"] == 0 and res["certified"] == res["capped"]
    assert res["first_certificate_iteration"] >= 20


def test_certificate_on_random_qps(fuzz_bin):
    res = json.loads(subprocess.run([fuzz_bin, "3000000"], capture_output=True, text=True, check=True).stdout)
    assert res["qps"] == 3000000 and res["mismatches"] == 0
    assert res["capped"] > 1000000 and res["certified"] > 0.95 * res["capped"]
