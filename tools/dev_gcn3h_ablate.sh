#!/bin/bash
# Dev (timing only, wrong numerics): where does the split16 graph-conv forward spend its time?  Builds variants of
# stgcn_gcn3h_fwd.hip from a patched COPY of stgcn_gcn3h_body.h (tmp_ab/abl/<name>/; the product sources are not
# touched), links each with the library's other objects into tmp_ab/lib_<name>.so; on the GPU box:
#   for v in base nolds nomfma novisit nodma nofence; do P2R_LIB_PATH=tmp_ab/lib_$v.so python tools/dev_split16_time.py gcn | head -1; done
#   base     the product kernel
#   nolds    gathers and coefficient reads come from registers (one v_mov each) instead of LDS
#   nomfma   the 12 MFMAs of a unit are not issued (their operands are kept alive)
#   novisit  the weight planes are loaded once per phase-0 pair instead of at every visit
#   nodma    no tile copies (LDS-DMA) in the main loop
#   nofence  no scheduling fences around the MFMA groups (the compiler may interleave)
#   mix      the residual of the operand split by v_fma_mixlo/hi_f16 (2 instructions per pair instead of 4; numerics intact)
#   noepi    no statistics / store epilogue (skipped at run time)
#   nocs     no combine and split stages (the gathered values are only waited for)
#   pure     nomfma + nolds + novisit + nodma: vector work, barriers and the epilogue only
#   sgb      no fences around the MFMA groups; a sched_group_barrier pipeline 12 x (1 MFMA, 3 VALU) over each unit's region
#   spread   the 12 MFMAs of a unit issued as 3 x 4 between the next unit's combine / split stages (numerics intact)
set -e
cd "$(dirname "$0")/.."
SRC=pose2room_amd/csrc
make -s -C $SRC >/dev/null
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize"
OTHERS=$(ls $SRC/*.o | grep -v stgcn_gcn3h_fwd.o)
for v in ${VARIANTS:-base nolds nomfma novisit nodma nofence spread mix noepi nocs pure sgb}; do
  d=tmp_ab/abl/$v; mkdir -p $d
  cp $SRC/stgcn_gcn3h_fwd.hip $SRC/gcn3h_sched_c.inc $d/
  python3 - "$v" $SRC/stgcn_gcn3h_body.h $d/stgcn_gcn3h_body.h $d/gcn3h_sched_c.inc <<'PY'
import sys, re
v, src, dst, inc = sys.argv[1:]
s = open(src).read()
def sub(old, new, count=1):
    global s
    assert old in s, old
    s = s.replace(old, new, count)
NOLDS = v in ('nolds', 'pure'); NOMFMA = v in ('nomfma', 'pure'); NOVISIT = v in ('novisit', 'pure'); NODMA = v in ('nodma', 'pure')
if v == 'noepi':
    sub("    if (p.stats) {\n      float *rs = rowstat", "    if (p.stats && p.total_tiles < 0) {\n      float *rs = rowstat")
    sub("    {\n      float *stg = lds + ((NPH - 1) & 1) * BUF;", "    if (p.total_tiles < 0) {\n      float *stg = lds + ((NPH - 1) & 1) * BUF;")
if v == 'nocs':
    sub("#define H3_C(k, ne, zero, h0, f0, h1, f1, h2, f2, h3, f3) h3_combine<ne, zero, h0, f0, h1, f1, h2, f2, h3, f3>(xv_[k], cf_[k], xagg);",
        "#define H3_C(k, ne, zero, h0, f0, h1, f1, h2, f2, h3, f3) { _Pragma(\"unroll\") for (int j_ = 0; j_ < ne; ++j_) asm volatile(\"\" :: \"v\"(xv_[k][j_][0]), \"v\"(xv_[k][j_][1]), \"v\"(xv_[k][j_][2]), \"v\"(xv_[k][j_][3]), \"v\"(cf_[k][j_])); }")
    a = s.index("#define H3_S(par)")
    b = s.index("#define H3_M(slot, par)")
    s = s[:a] + "#define H3_S(par) { }\n" + s[b:]
    sub("  p2r_h8 b1_[2], b2_[2];", "  p2r_h8 b1_[2], b2_[2];\n  for (int i_ = 0; i_ < 8; ++i_) { b1_[0][i_] = (_Float16)(lane + i_); b1_[1][i_] = (_Float16)(lane - i_); b2_[0][i_] = (_Float16)(i_ * 0.5f); b2_[1][i_] = (_Float16)(i_ * 0.25f); }\n  asm volatile(\"\" : \"+v\"(b1_[0]), \"+v\"(b1_[1]), \"+v\"(b2_[0]), \"+v\"(b2_[1]));")
if NOLDS:
    sub("xv[j][i] = *reinterpret_cast<const float *>(xl + off[j] + i * (4 * H3_RS * 4));",
        "{ float t_ = __builtin_bit_cast(float, (unsigned)(size_t)xl); asm volatile(\"\" : \"+v\"(t_)); xv[j][i] = t_; }")
    sub("c[j] = *reinterpret_cast<const float *>(cl + 4 * ci[j]);",
        "{ float t_ = __builtin_bit_cast(float, (unsigned)(size_t)cl); asm volatile(\"\" : \"+v\"(t_)); c[j] = t_; }")
if NOMFMA:
    a = s.index("__device__ __forceinline__ void h3_mfma12(")
    b = s.index("// x = x1 + x2 for a pair of values")
    s = s[:a] + '''__device__ __forceinline__ void h3_mfma12(f32x4 (&acc)[4], const p2r_h8 (&a1)[4], const p2r_h8 (&a2)[4], const p2r_h8 &b1,
                                          const p2r_h8 &b2) {
  asm volatile("" :: "v"(b1), "v"(b2), "v"(a1[0]), "v"(a2[0]), "v"(a1[1]), "v"(a2[1]), "v"(a1[2]), "v"(a2[2]), "v"(a1[3]), "v"(a2[3]));
}

''' + s[b:]
if NOVISIT:
    sub("#define H3_VISIT(pair) { load_a(aS, pair, ph); }", "#define H3_VISIT(pair) { }")
    sub("    load_a(aS, pair0, (ph + 1) & (H3_NPH - 1));                                           \\\n", "")
if NODMA:
    sub("#define H3_PIECE(piece) { if (copy) dma_piece(piece); }", "#define H3_PIECE(piece) { }")
    sub("    if (copy) { _Pragma(\"unroll\") for (int i_ = pieces; i_ < H3_PW; ++i_) dma_piece(i_); } \\\n", "")
if v == 'nofence':
    sub("    __builtin_amdgcn_sched_barrier(0);                        \\\n    h3_mfma12(acc[slot], aS[0], aS[1], b1_[par], b2_[par]);   \\\n    __builtin_amdgcn_sched_barrier(0);                        \\\n",
        "    h3_mfma12(acc[slot], aS[0], aS[1], b1_[par], b2_[par]);   \\\n")
if v == 'mix':
    a = s.index("#define H3_S(par)")
    b = s.index("#define H3_M(slot, par)")
    s = s[:a] + """#define H3_S(par)                                                                         \\
  {                                                                                       \\
    unsigned p_[4], r_[4];                                                                \\
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                    \\
      p2r_f2 x_ = {xagg[2 * q_], xagg[2 * q_ + 1]};                                       \\
      asm volatile("" : "+v"(x_));                                                        \\
      p_[q_] = __builtin_bit_cast(unsigned, __builtin_convertvector(x_, p2r_h2));         \\
    }                                                                                     \\
    asm("v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel_hi:[1,0,0]\\n\\tv_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel_hi:[1,0,0]\\n\\t" \\
        "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel_hi:[1,0,0]\\n\\tv_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel_hi:[1,0,0]\\n\\t" \\
        "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\\n\\tv_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\\n\\t" \\
        "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\\n\\tv_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\\n\\ts_nop 1" \\
        : "=&v"(r_[0]), "=&v"(r_[1]), "=&v"(r_[2]), "=&v"(r_[3])                          \\
        : "v"(p_[0]), "v"(p_[1]), "v"(p_[2]), "v"(p_[3]), "v"(xagg[0]), "v"(xagg[1]), "v"(xagg[2]), "v"(xagg[3]), \\
          "v"(xagg[4]), "v"(xagg[5]), "v"(xagg[6]), "v"(xagg[7]));                        \\
    b1_[par] = __builtin_bit_cast(p2r_h8, h3_u4{p_[0], p_[1], p_[2], p_[3]});              \\
    b2_[par] = __builtin_bit_cast(p2r_h8, h3_u4{r_[0], r_[1], r_[2], r_[3]});              \\
  }
""" + s[b:]
if v == 'sgb':
    sub("    __builtin_amdgcn_sched_barrier(0);                        \\\n    h3_mfma12(acc[slot], aS[0], aS[1], b1_[par], b2_[par]);   \\\n    __builtin_amdgcn_sched_barrier(0);                        \\\n",
        "    h3_mfma12(acc[slot], aS[0], aS[1], b1_[par], b2_[par]);   \\\n")
    sub("    b2_[par] = __builtin_bit_cast(p2r_h8, h3_u4{r_[0], r_[1], r_[2], r_[3]});              \\\n",
        "    b2_[par] = __builtin_bit_cast(p2r_h8, h3_u4{r_[0], r_[1], r_[2], r_[3]});              \\\n"
        "    _Pragma(\"unroll\") for (int i_ = 0; i_ < 12; ++i_) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, SGB_N, 0); } \\\n"
        "    __builtin_amdgcn_sched_barrier(0);                                                    \\\n")
    s = "#define SGB_N 3\n" + s
if v == 'spread':
    sub("#define H3_END(pieces, pair0)", """#define H3_MX(part, slot, par)                                \\
  {                                                           \\
    __builtin_amdgcn_sched_barrier(0);                        \\
    h3_mfma4<part>(acc[slot], aS[0], aS[1], b1_[par], b2_[par]); \\
    __builtin_amdgcn_sched_barrier(0);                        \\
  }
#define H3_END(pieces, pair0)""")
    sub("// x = x1 + x2 for a pair of values", """template <int PART>
__device__ __forceinline__ void h3_mfma4(f32x4 (&acc)[4], const p2r_h8 (&a1)[4], const p2r_h8 (&a2)[4], const p2r_h8 &b1,
                                         const p2r_h8 &b2) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (PART == 0) {
#ifdef H3_RES_SCALED
      const p2r_h8 a1s = a1[m] * (_Float16)(1.0 / P2R_RES_SCALE);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1s, b2, acc[m], 0, 0, 0);
#else
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[m], b2, acc[m], 0, 0, 0);
#endif
    } else if (PART == 1) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[m], b1, acc[m], 0, 0, 0);
    } else {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[m], b1, acc[m], 0, 0, 0);
    }
  }
}

// x = x1 + x2 for a pair of values""")
    t = open(inc).read().split('\n')
    out, i = [], 0
    while i < len(t):
        m = re.match(r'\s*H3_M\((\d+), (\d+)\)', t[i])
        if not m:
            out.append(t[i]); i += 1; continue
        sl, pr = m.group(1), m.group(2)
        j = i + 1
        piece = visit = None
        if 'H3_PIECE' in t[j]: piece = t[j]; j += 1
        if 'H3_VISIT' in t[j]: visit = t[j]; j += 1
        mk = lambda part: '  H3_MX(%d, %s, %s) \\' % (part, sl, pr)
        out.append(mk(0))
        if piece: out.append(piece)
        if 'H3_C(' in t[j]:
            out.append(t[j]); j += 1
            out.append(mk(1))
            while 'H3_G(' in t[j] and 'H3_C(' in t[j + 1] and 'H3_S' not in t[j + 2].split('(')[0] + 'x' and not re.match(r'\s*H3_(S|M)', t[j + 2]) :
                out += [t[j], t[j + 1]]; j += 2
            # remaining (G, C) pairs of the next unit's later chunks
            while 'H3_G(' in t[j] and 'H3_C(' in t[j + 1]:
                out += [t[j], t[j + 1]]; j += 2
            out.append(mk(2))
        else:
            out += [mk(1), mk(2)]
        if visit: out.append(visit)
        i = j
    open(inc, 'w').write('\n'.join(out))
open(dst, 'w').write(s)
PY
  /opt/rocm/bin/hipcc $FLAGS -I $SRC -I include -c $d/stgcn_gcn3h_fwd.hip -o $d/fwd.o -save-temps=obj 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tmp_ab/lib_$v.so $OTHERS $d/fwd.o
  echo "$v: $(grep -E '^\s+\.vgpr_spill_count' $d/stgcn_gcn3h_fwd-hip-amdgcn-amd-amdhsa-gfx950.s | head -1)"
done
