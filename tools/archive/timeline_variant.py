#!/usr/bin/env python3
"""tools/timeline_variant.py — builds melonix_amd/lib/variants/timeline.so: the shipped STFT kernels with s_memtime stamps at
the phase boundaries of two consecutive frames of ONE workgroup (CPU container; run tools/timeline_run.py on the GPU box).

The product sources are not touched: the stamps are textual insertions into a temporary copy (like tools/ab_variant.sh).
Stamps sit only where the wave's LDS queue is empty anyway (behind barriers, behind the hand-placed lgkmcnt(0) of the
gathers, in front of a scatter) — s_memtime is a scalar memory operation and its result needs lgkmcnt(0), which would
otherwise drain the very LDS traffic the timeline is about.  The stamp buffer travels in StftArgs.rgb (unused by the bulk
kernels), the (workgroup, frame) to stamp in StftArgs.ranges; both come from the environment at launch (MX_TL_BUF,
MX_TL_BLOCK, MX_TL_FRAME)."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def patch(text, old, new, count=1):
    assert text.count(old) >= 1, old
    return text.replace(old, new, count)


def main():
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "melonix_amd"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    shutil.copytree(os.path.join(ROOT, "melonix_amd", "csrc"), os.path.join(tmp, "melonix_amd", "csrc"))
    for f in os.listdir(os.path.join(ROOT, "melonix_amd")):
        if f.endswith(".py"):
            shutil.copy(os.path.join(ROOT, "melonix_amd", f), os.path.join(tmp, "melonix_amd", f))
    os.makedirs(os.path.join(tmp, "melonix_amd", "build"))
    for f in os.listdir(os.path.join(ROOT, "melonix_amd", "build")):
        if f.endswith(".o") and not f.startswith(("stft_kernels", "capi_ctx")):
            shutil.copy(os.path.join(ROOT, "melonix_amd", "build", f), os.path.join(tmp, "melonix_amd", "build", f))

    p = os.path.join(tmp, "melonix_amd", "csrc", "stft_kernel_impl.h")
    s = open(p).read()
    s = patch(s, "  for (int64_t f = f0; f < f1; ++f) {\n    // Everything below that depends only on the thread index",
              "  unsigned long long *const tl_buf = reinterpret_cast<unsigned long long *>(a.rgb);\n"
              "  const unsigned tl_sel = (unsigned)(uintptr_t)a.ranges;\n"
              "#define MX_STAMP(i) do { const int64_t df_ = (f - f0) - (int64_t)(tl_sel & 255u); if (!CMAP && MODE != kRanges && tl_buf && lb == (tl_sel >> 8) && (df_ == 0 || df_ == 1)) { "
              "const unsigned long long tm_ = __builtin_amdgcn_s_memtime(); if ((t_ & 63) == 0) tl_buf[((int)df_ * 13 + (i)) * NW + (t_ >> 6)] = tm_; } } while (0)\n"
              "  for (int64_t f = f0; f < f1; ++f) {\n    MX_STAMP(0);\n    // Everything below that depends only on the thread index")
    s = patch(s, "\n    cpx v[P::E];\n    pass1<P>(Y, v);\n", "\n    MX_STAMP(1);\n    cpx v[P::E];\n    pass1<P>(Y, v);\n    MX_STAMP(2);\n")
    s = patch(s, "    store_t1<P>(t, v, lds);\n    MX_BARRIER();\n", "    store_t1<P>(t, v, lds);\n    MX_BARRIER();\n    MX_STAMP(3);\n")
    s = patch(s, "      load_t1_tw2<P>(t, v, lds, ltw2, w2);\n", "      load_t1_tw2<P>(t, v, lds, ltw2, w2);\n      MX_STAMP(4);\n")
    s = patch(s, "      MX_BARRIER();\n      pass2_reg<P>(v, w2);\n", "      MX_BARRIER();\n      MX_STAMP(5);\n      pass2_reg<P>(v, w2);\n      MX_STAMP(6);\n")
    s = patch(s, "      load_t1<P>(t, v, lds);\n", "      load_t1<P>(t, v, lds);\n      MX_STAMP(4);\n")
    s = patch(s, "      MX_BARRIER();\n      pass2<P>(t, v, ltw2);\n", "      MX_BARRIER();\n      MX_STAMP(5);\n      pass2<P>(t, v, ltw2);\n      MX_STAMP(6);\n")
    s = patch(s, "    store_t2<P>(t, v, lds);\n    MX_BARRIER();\n    load_t2<P>(t, v, lds);\n    MX_BARRIER();",
              "    store_t2<P>(t, v, lds);\n    MX_BARRIER();\n    MX_STAMP(7);\n    load_t2<P>(t, v, lds);\n    MX_STAMP(8);\n    MX_BARRIER();\n    MX_STAMP(9);")
    s = patch(s, "    if constexpr (PREFETCH && !kSlide) {\n      // the transform's registers are free again",
              "    MX_STAMP(10);\n    if constexpr (PREFETCH && !kSlide) {\n      // the transform's registers are free again")
    s = patch(s, "    // ---- magnitudes ----\n", "    MX_STAMP(11);\n    // ---- magnitudes ----\n")
    s = patch(s, "    if constexpr (!DEFER && !DIRECT) {\n      if (want_rows) {\n        MX_BARRIER();",
              "    MX_STAMP(12);\n    if constexpr (!DEFER && !DIRECT) {\n      if (want_rows) {\n        MX_BARRIER();")
    open(p, "w").write(s)

    p = os.path.join(tmp, "melonix_amd", "csrc", "capi_ctx.cpp")  # stft_launch lives in the context unit
    s = open(p).read()
    s = patch(s, "  s.rgb = d_rgb;",
              "  s.rgb = d_rgb;\n"
              "  if (!d_rgb && mode != kRanges && getenv(\"MX_TL_BUF\")) {\n"
              "    s.rgb = reinterpret_cast<uint8_t *>((uintptr_t)strtoull(getenv(\"MX_TL_BUF\"), nullptr, 0));\n"
              "    const unsigned blk = getenv(\"MX_TL_BLOCK\") ? (unsigned)atoi(getenv(\"MX_TL_BLOCK\")) : 0u;\n"
              "    const unsigned frm = getenv(\"MX_TL_FRAME\") ? (unsigned)atoi(getenv(\"MX_TL_FRAME\")) : 0u;\n"
              "    s.ranges = reinterpret_cast<const int32_t *>((uintptr_t)((blk << 8) | (frm & 255u)));\n"
              "  }")
    open(p, "w").write(s)

    subprocess.check_call([sys.executable, "-c", "import melonix_amd.build as b; b.build()"], cwd=tmp)
    out = os.path.join(ROOT, "melonix_amd", "lib", "variants")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(tmp, "melonix_amd", "lib", "libmelonix_amd.so"), os.path.join(out, "timeline.so"))
    shutil.rmtree(tmp)
    print("built", os.path.join(out, "timeline.so"))


if __name__ == "__main__":
    main()
