"""tests/tools/pv_diag2.py — debugging aid (GPU box): dumps the staged phase-vocoder job's intermediate rows (magnitudes, phase words,
peak maps, synthesis phases) through mx_debug_pv_row — which only exists in a library built with -DMX_PV_DEBUG
(melonix_amd.build.build(force=True, extra_defines=["-DMX_PV_DEBUG"])) — and compares them with oracle/pv_oracle.py frame by frame."""
import os, sys, ctypes as C
import numpy as np
import torch
torch.cuda.init()
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import melonix_amd as mx
from melonix_amd import _capi
from oracle import pv_oracle as pv
from conftest import accum_sweep, SR
ctx = mx.Context(0)
w = accum_sweep(3 * SR)
a = ctx.upload(w)
st = 3.0
r = pv.ratio(st); F, ap = pv.plan(len(w), r)
mags, ph = pv.analysis(w.astype(np.float64), ap)
act = mags >= np.float32(pv.ACTIVE_REL) * mags.max(axis=1, keepdims=True)
Phi = pv.synthesis_phases(ph, ap, mags)
sums, org = ctx.pv_shard_analyze(a, st, 0, 1)
L = _capi.lib()
L.mx_debug_pv_row.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
def row(kind, f):
    out = np.empty(64 if kind == 2 else 2048, dtype=[np.float32, np.uint32, np.uint32, np.uint32][kind])
    assert L.mx_debug_pv_row(ctx.handle, kind, f, out.ctypes.data) == 0
    return out
for f in range(0, 12):
    gm, gp, gmap = row(0, f), row(1, f), row(2, f)
    gpk = ((gmap[:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).reshape(-1)
    opk = pv.peaks(mags[f], act[f])
    gact = (gp & 1).astype(bool)
    nd = np.flatnonzero(gpk != opk)
    print(f"frame {f}: act diff {np.count_nonzero(gact != act[f])}, peak diff {len(nd)} at {nd[:8]}; mag relerr {np.abs(gm-mags[f]).max()/mags[f].max():.1e}")
head, tail = ctx.pv_shard_synthesize(None)
for f in range(0, 40):
    gphi = row(3, f)
    d = (gphi.astype(np.int64) - Phi[f].astype(np.int64) + 2**31) % 2**32 - 2**31
    werr = mags[f] * np.abs(np.exp(2j*np.pi*d/2**32) - 1)
    top = np.argsort(werr)[::-1][:4]
    print(f"frame {f}: weighted phase err max {werr.max():.2e} (frame peak {mags[f].max():.2e}) at bins {top} mags rel {mags[f][top]/mags[f].max()} act {act[f][top]} turns {d[top]/2**32}")
f32, _ = ctx.pv_shard_finish(len(w), None, None, True, False)
ref = pv.pitch_shift(w.astype(np.float64), st)
err = np.abs(f32 - ref); print("staged out err max", err.max(), err.argmax(), "first block", err[:2400].max())
g32, _ = ctx.pv_pitch_shift(a, st)
err = np.abs(g32 - ref); print("direct out err max", err.max(), err.argmax())
print("staged vs direct", np.abs(g32 - f32).max())
s = pv.synthesis(mags, Phi)
print("oracle s around start:", s[2048-3:2048+6])
# isolate synthesis: the oracle's synthesis fed with the GPU's rows (re-run the staged stages to have the rows again)
ctx.pv_shard_analyze(a, st, 0, 1); ctx.pv_shard_synthesize(None)
FF = 64
gm = np.stack([row(0, f) for f in range(FF)]).astype(np.float64)
gph = np.stack([row(3, f) for f in range(FF)])
s_g = pv.synthesis(gm, gph)
s_o = pv.synthesis(mags[:FF], Phi[:FF])
pos = np.arange(1000, dtype=np.float64) * r + pv.N // 2
m = np.floor(pos).astype(np.int64); tt = pos - m
og = (1 - tt) * s_g[m] + tt * s_g[m + 1]
oo = (1 - tt) * s_o[m] + tt * s_o[m + 1]
print("oracle-synth(GPU rows) vs ref:", np.abs(og - ref[:1000]).max(), " oracle-synth(oracle rows) vs ref:", np.abs(oo - ref[:1000]).max())
print("GPU out vs oracle-synth(GPU rows):", np.abs(g32[:1000] - og).max())
d = (gph.astype(np.int64) - Phi[:FF].astype(np.int64) + 2**31) % 2**32 - 2**31
werr = mags[:FF] * np.abs(np.exp(2j*np.pi*d/2**32) - 1)
print("sum of weighted phase errors per frame:", werr.sum(axis=1)[:12])
print("nan in gpu mags", np.isnan(gm).sum(), "mag abs diff max", np.abs(gm - mags[:FF]).max())
e = np.abs(g32[:1000] - og)
print("err per 50 samples:", " ".join(f"{x:.0e}" for x in e.reshape(-1, 50).max(axis=1)))
print("oracle-synth(GPU rows) vs ref:", np.abs(og - ref[:1000]).max())
# does the error come from frames beyond FF? use all rows
