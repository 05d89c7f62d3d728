#!/usr/bin/env python3
"""tools/timeline_pv.py build | run [sweep|rich] — per-wave phase timeline of pv_analysis and pv_synthesis.

build (CPU container): melonix_amd/lib/variants/timeline_pv.so = the shipped phase-vocoder kernels with s_memtime stamps at
  the phase boundaries of two consecutive frames of ONE workgroup.  The product sources are not touched: the stamps are
  textual insertions into a temporary copy (as tools/timeline_variant.py does for the STFT kernels).  A stamp waits for
  lgkmcnt(0) (s_memtime is a scalar memory operation), so a stamp behind a phase that issued LDS traffic includes the time
  that traffic takes to complete — which is what the phase costs the wave anyway when a barrier follows.
run (GPU box): the 60-minute +3 st call; a few (workgroup, frame) samples per kernel; mean ticks per phase and wave
  (s_memtime ticks at the shader clock on this part)."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "melonix_amd", "lib", "variants", "timeline_pv.so")

ANALYSIS = ["window + pass 1", "barrier", "gathers issued, T1 scatter", "barrier", "T1 gather + twiddles", "barrier", "pass 2 + T2 scatter",
            "barrier", "T2 gather", "barrier", "pass 3 + split", "X staged, frame max", "barrier", "next samples + row stores issued",
            "peak search", "barrier", "records (wave 1 first)", "peak numbering (wave 0)", "-> next frame top"]
SYNTHESIS = ["row + offsets read (LDS), phasors, pre-split, pass 1", "barrier", "next row requested (LDS-DMA), T1 scatter, offsets zeroed", "barrier", "T1 gather", "barrier",
             "twiddles, pass 2, T2 scatter, offsets filled", "next row landed (vmcnt 0)", "barrier", "T2 gather", "pass 3",
             "overlap-add (registers), hop out", "-> next frame top"]


def patch(text, old, new, count=1):
    assert text.count(old) >= 1, old
    return text.replace(old, new, count)


def build():
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "melonix_amd"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    shutil.copytree(os.path.join(ROOT, "melonix_amd", "csrc"), os.path.join(tmp, "melonix_amd", "csrc"))
    for f in os.listdir(os.path.join(ROOT, "melonix_amd")):
        if f.endswith(".py"):
            shutil.copy(os.path.join(ROOT, "melonix_amd", f), os.path.join(tmp, "melonix_amd", f))
    os.makedirs(os.path.join(tmp, "melonix_amd", "build"))
    for f in os.listdir(os.path.join(ROOT, "melonix_amd", "build")):
        if f.endswith(".o") and not f.startswith("pv_kernels"):
            shutil.copy(os.path.join(ROOT, "melonix_amd", "build", f), os.path.join(tmp, "melonix_amd", "build", f))

    p = os.path.join(tmp, "melonix_amd", "csrc", "pv_kernels.hip")
    s = open(p).read()
    s = patch(s, "__global__ __launch_bounds__(PV::T) void pv_analysis(const PvArgs a) {",
              "__device__ unsigned long long *mx_tl_buf;\n__device__ unsigned mx_tl_sel[4];  // analysis block, frame; synthesis block, frame\n"
              "#define MX_STAMP(K, NST, i) do { const int64_t df_ = (f - tl_f0) - tl_frm; if (tl_on && (df_ == 0 || df_ == 1)) { "
              "const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if ((t_ & 63) == 0) tl_ptr[(K) * 256 + ((int)df_ * (NST) + (i)) * 2 + (t_ >> 6)] = tm_; } } while (0)\n"
              "__global__ __launch_bounds__(PV::T) void pv_analysis(const PvArgs a) {")
    # ---- analysis ----
    s = patch(s, "  int cur = 0;\n  for (int64_t f = f0; f < f1; ++f) {\n",
              "  int cur = 0;\n  unsigned long long *const tl_ptr = mx_tl_buf;\n  const bool tl_on = tl_ptr != nullptr && lb == mx_tl_sel[0];\n  const int64_t tl_f0 = f0, tl_frm = (int64_t)mx_tl_sel[1];\n"
              "  for (int64_t f = f0; f < f1; ++f) {\n    MX_STAMP(0, 19, 0);\n")
    s = patch(s, "    pass1<P>(Y, v);\n    __syncthreads();  // every wave is past the previous frame's peak numbering: plist and npk are complete\n",
              "    pass1<P>(Y, v);\n    MX_STAMP(0, 19, 1);\n    __syncthreads();\n    MX_STAMP(0, 19, 2);\n")
    s = patch(s, "    store_t1<P>(t, v, lds);\n    __syncthreads();\n    cpx w2[P::R2 - 1];\n    load_t1_tw2<P>(t, v, lds, ltw2, w2);\n    __syncthreads();\n    pass2_reg<P>(v, w2);\n    store_t2<P>(t, v, lds);\n    __syncthreads();\n    load_t2<P>(t, v, lds);\n    __syncthreads();  // every wave has its T2 read: the image is free for X_f\n",
              "    store_t1<P>(t, v, lds);\n    MX_STAMP(0, 19, 3);\n    __syncthreads();\n    MX_STAMP(0, 19, 4);\n    cpx w2[P::R2 - 1];\n    load_t1_tw2<P>(t, v, lds, ltw2, w2);\n    MX_STAMP(0, 19, 5);\n    __syncthreads();\n    MX_STAMP(0, 19, 6);\n"
              "    pass2_reg<P>(v, w2);\n    store_t2<P>(t, v, lds);\n    MX_STAMP(0, 19, 7);\n    __syncthreads();\n    MX_STAMP(0, 19, 8);\n    load_t2<P>(t, v, lds);\n    MX_STAMP(0, 19, 9);\n    __syncthreads();\n    MX_STAMP(0, 19, 10);\n")
    s = patch(s, "    float mx2 = 0.f;\n#pragma unroll\n    for (int o = 0; o < P::E; ++o) {\n      const float n2 = cnorm2(X[o]);",
              "    MX_STAMP(0, 19, 11);\n    float mx2 = 0.f;\n#pragma unroll\n    for (int o = 0; o < P::E; ++o) {\n      const float n2 = cnorm2(X[o]);")
    s = patch(s, "    if (t < W) pkb[m0][t + 1] = 0u;\n    __syncthreads();\n",
              "    if (t < W) pkb[m0][t + 1] = 0u;\n    MX_STAMP(0, 19, 12);\n    __syncthreads();\n    MX_STAMP(0, 19, 13);\n")
    s = patch(s, "    const float thr2 = kPvActiveRel2 * (red[cur][0] > red[cur][1] ? red[cur][0] : red[cur][1]);\n",
              "    MX_STAMP(0, 19, 14);\n    const float thr2 = kPvActiveRel2 * (red[cur][0] > red[cur][1] ? red[cur][0] : red[cur][1]);\n")
    s = patch(s, "    __syncthreads();  // this frame's peak map is complete\n",
              "    MX_STAMP(0, 19, 15);\n    __syncthreads();\n    MX_STAMP(0, 19, 16);\n")
    s = patch(s, "    // the first wavefront numbers this frame's peaks",
              "    MX_STAMP(0, 19, 17);\n    // the first wavefront numbers this frame's peaks")
    s = patch(s, "    pend = f > f0;  // (the first frame's records are pv_heads')\n", "    MX_STAMP(0, 19, 18);\n    pend = f > f0;\n")
    # ---- synthesis ----
    s = patch(s, "  for (int64_t f = f0; f < f1; ++f) {\n    const int t = t_;\n    // (the sixteen products",
              "  unsigned long long *const tl_ptr = mx_tl_buf;\n  const bool tl_on = tl_ptr != nullptr && blk == (int64_t)mx_tl_sel[2];\n  const int64_t tl_f0 = f0, tl_frm = (int64_t)mx_tl_sel[3];\n"
              "  for (int64_t f = f0; f < f1; ++f) {\n    MX_STAMP(1, NSY, 0);\n    const int t = t_;\n    // (the sixteen products")
    s = patch(s, "    pass1<P>(Y, v);\n    __syncthreads();  // (every wave has read this frame's row and offsets, and its T2 columns of the previous frame)\n    if (f + 1 < f1) request_row(f + 1, t);\n    store_t1<P>(t, v, lds);\n    zero_cd(t);\n    __syncthreads();\n    load_t1<P>(t, v, lds);\n    __syncthreads();\n",
              "    pass1<P>(Y, v);\n    MX_STAMP(1, NSY, 1);\n    __syncthreads();\n    MX_STAMP(1, NSY, 2);\n    if (f + 1 < f1) request_row(f + 1, t);\n    store_t1<P>(t, v, lds);\n    zero_cd(t);\n    MX_STAMP(1, NSY, 3);\n    __syncthreads();\n    MX_STAMP(1, NSY, 4);\n    load_t1<P>(t, v, lds);\n    MX_STAMP(1, NSY, 5);\n    __syncthreads();\n    MX_STAMP(1, NSY, 6);\n")
    s = patch(s, "    cnt1 = cnt2;\n    row_landed();  // (requested three barriers ago)\n    __syncthreads();\n", "    cnt1 = cnt2;\n    MX_STAMP(1, NSY, 7);\n    row_landed();\n    MX_STAMP(1, NSY, 8);\n    __syncthreads();\n    MX_STAMP(1, NSY, 9);\n")
    s = patch(s, "    pass3_col(g3p, 0);\n    pass3_col(g3q, P::R3);\n",
              "    MX_STAMP(1, NSY, 10);\n    pass3_col(g3p, 0);\n    pass3_col(g3q, P::R3);\n    MX_STAMP(1, NSY, 11);\n")
    s = patch(s, "        reinterpret_cast<float2 *>(a.halo + (size_t)blk * kPvHalo + (f - f0) * kPvHs)[t] = make_float2(hopv.x, hopv.y);\n      }\n    }\n",
              "        reinterpret_cast<float2 *>(a.halo + (size_t)blk * kPvHalo + (f - f0) * kPvHs)[t] = make_float2(hopv.x, hopv.y);\n      }\n    }\n    MX_STAMP(1, NSY, 12);\n")
    s = s.replace("NSY", str(len(SYNTHESIS)))
    # ---- the stamp buffer and the (workgroup, frame) selection come from the environment at launch ----
    s = patch(s, "hipError_t launch_pv(const PvArgs &a, hipStream_t s) {\n  if (a.frames <= 0 || a.n <= 0) return hipSuccess;\n",
              "hipError_t launch_pv(const PvArgs &a, hipStream_t s) {\n  if (a.frames <= 0 || a.n <= 0) return hipSuccess;\n"
              "  {\n    unsigned long long *b = getenv(\"MX_TL_BUF\") ? reinterpret_cast<unsigned long long *>((uintptr_t)strtoull(getenv(\"MX_TL_BUF\"), nullptr, 0)) : nullptr;\n"
              "    unsigned sel[4] = {0, 0, 0, 0};\n    if (getenv(\"MX_TL_SEL\")) sscanf(getenv(\"MX_TL_SEL\"), \"%u,%u,%u,%u\", &sel[0], &sel[1], &sel[2], &sel[3]);\n"
              "    hipMemcpyToSymbolAsync(HIP_SYMBOL(mx_tl_buf), &b, sizeof(b), 0, hipMemcpyHostToDevice, s);\n"
              "    hipMemcpyToSymbolAsync(HIP_SYMBOL(mx_tl_sel), sel, sizeof(sel), 0, hipMemcpyHostToDevice, s);\n    hipStreamSynchronize(s);\n  }\n")
    if "#include <cstdlib>" not in s:
        s = "#include <cstdio>\n#include <cstdlib>\n" + s
    open(p, "w").write(s)
    subprocess.check_call([sys.executable, "-c", "import melonix_amd.build as b; b.build()"], cwd=tmp)
    os.makedirs(os.path.dirname(VARIANT), exist_ok=True)
    shutil.copy(os.path.join(tmp, "melonix_amd", "lib", "libmelonix_amd.so"), VARIANT)
    shutil.rmtree(tmp)
    print("built", VARIANT)


def run(signal):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch

    import melonix_amd as mx

    mx._capi.LIB_PATH = os.environ.get("MX_TL_LIB", VARIANT)
    from bench import SR, gen_shard

    dev = torch.device("cuda", 0)
    n = 60 * 60 * SR
    audio_t = gen_shard(torch, dev, 0, 1, n, mx.MX_AUDIO_PAD)
    if signal == "rich":
        from bench import add_noise

        pad = mx.MX_AUDIO_PAD
        T = n / SR
        chunk = 1 << 24
        for c in range(0, n, chunk):
            m = min(chunk, n - c)
            t = torch.arange(c, c + m, dtype=torch.float64, device=dev) / SR
            ph = 110.0 * t + (1760.0 - 110.0) * t * t / (2 * T)
            acc = torch.zeros(m, dtype=torch.float64, device=dev)
            for h in range(2, 13):
                acc += (0.25 / h) * torch.sin(2 * np.pi * h * ph)
            audio_t[pad + c:pad + c + m] = (audio_t[pad + c:pad + c + m].to(torch.float64) * 0.5 + acc).to(torch.float32)
        add_noise(torch, dev, audio_t, 0, 1, n, pad)
    ctx = mx.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    audio = ctx.wrap_device(audio_t.data_ptr(), n, keepalive=audio_t)
    f32 = torch.empty(n, dtype=torch.float32, device=dev)
    buf = torch.zeros(512, dtype=torch.int64, device=dev)
    os.environ["MX_TL_BUF"] = str(buf.data_ptr())
    frames = int(n * 2.0 ** (3 / 12.0) / 256)
    acc = {0: [], 1: []}
    for ablk, sblk in ((frames // 16 // 7, frames // 32 // 7), (frames // 16 // 3, frames // 32 // 3), (frames // 16 // 2 + 11, frames // 32 // 2 + 11),
                       (frames // 16 - 300, frames // 32 - 150)):
        for afrm, sfrm in ((2, 3), (8, 15), (14, 29)):
            os.environ["MX_TL_SEL"] = f"{ablk},{afrm},{sblk},{sfrm}"
            buf.zero_()
            for _ in range(2):
                ctx.pv_pitch_shift_dev(audio, 3.0, f32.data_ptr(), 0)
            torch.cuda.synchronize()
            tl = buf.cpu().numpy().astype(np.int64)
            for k, nst in ((0, len(ANALYSIS)), (1, len(SYNTHESIS))):
                x = tl[256 * k:256 * k + 2 * nst * 2].reshape(2, nst, 2)
                if (x == 0).any():
                    continue
                d = np.diff(x[0], axis=0)                       # [nst-1][2 waves]
                last = (x[1, 0] - x[0, nst - 1])[None, :]       # next frame's top
                acc[k].append(np.vstack([d, last]))
    for k, names, title in ((0, ANALYSIS, "pv_analysis"), (1, SYNTHESIS, "pv_synthesis")):
        if not acc[k]:
            print(f"== {title}: no stamps came back")
            continue
        A = np.stack(acc[k]).astype(np.float64)
        tot = A.sum(axis=1)
        print(f"== {title} ({signal}): {len(acc[k])} (workgroup, frame) samples; a frame = {tot.mean():.0f} ticks per wave (min {tot.min():.0f}, max {tot.max():.0f}); "
              f"the {len(names)} stamps cost ~60-100 ticks each")
        mean = A.mean(axis=0)
        for i, nm in enumerate(names):
            print(f"   {nm:48s} " + " ".join(f"{x:6.0f}" for x in mean[i]) + f"   | mean {mean[i].mean():6.0f} = {100 * mean[i].mean() / tot.mean():5.1f} %")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2] if len(sys.argv) > 2 else "sweep")
