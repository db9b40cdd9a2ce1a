"""Round 6: the planes of a LayerNorm forward leave through LDS in big launches (csrc/layernorm.hip STAGE: whole-line stores instead
of 64-byte row segments).  The staged launch must write EXACTLY what the unstaged one writes -- fp32 outputs and every byte of the
plane buffer that belongs to a row -- for both plane formats, the gather site and the residual site, row counts that are and are not
multiples of four, D = 256 .. 1024.  The row threshold is read once per process: each policy runs in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, hashlib
sys.path.insert(0, %r)
import torch
from pixelrec_amd import ops
torch.manual_seed(11)
dev = "cuda"
out = []
for (B, L, D) in ((7, 50, 512), (33, 10, 256), (5, 13, 768), (64, 50, 512), (9, 7, 1024)):
    table, pos = torch.randn(500, D, device=dev), torch.randn(L, D, device=dev)
    gamma, beta = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
    idx = torch.randint(0, 500, (B, L), device=dev)
    x, res = torch.randn(B, L, D, device=dev), torch.randn(B, L, D, device=dev)
    for fmt in (True, "h2"):
        y, xh, rs, yp = ops.input_ln_fwd(table, idx, L, B, L, pos, gamma, beta, 1e-12, 0.1, 5, 0, planes=fmt)
        y2, xh2, rs2, yp2 = ops.ln_residual_fwd(x, res, gamma, beta, 1e-12, 0.2, 7, 3, planes=fmt)
        torch.cuda.synchronize()
        for t in (y, xh, rs, yp.to_dense(), y2, xh2, rs2, yp2.to_dense()):
            out.append(hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest())
        assert torch.equal(yp.to_dense(), y.view(B * L, D)) or fmt == "h2"
ops.raise_on_bad_indices("cuda")
print("HASHES " + " ".join(out))
""" % ROOT


def _run(stage_rows):
    env = dict(os.environ, PXR_LN_STAGE_ROWS=str(stage_rows))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("HASHES ")][-1]
    return line.split()[1:]


def test_staged_plane_stores_write_the_same_bytes():
    staged, plain = _run(1), _run(1 << 30)
    assert len(staged) == len(plain) == 5 * 2 * 8
    assert staged == plain
