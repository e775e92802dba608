#!/usr/bin/env python3
"""Generates issue_rules.hip: micro-benchmarks of the gfx950 VALU issue rules that decide how the RK step should be ORDERED.

Every kernel is one asm block: a loop whose body is a fixed instruction pattern (no compiler scheduling in between).  The host
times each pattern on the whole chip at 1 / 2 / 4 / 6 / 8 waves per SIMD and prints cycles per wave-instruction per SIMD at a
nominal 2.4 GHz.  What the patterns ask:
  dK          v_fma_f32 chains, an instruction depends on the one K places before it (K = 1..8)
  litK        the same with v_fmac_f32 and a 32-bit literal (the Cash-Karp stages' encoding)
  aabb / aaabbb  dependent neighbours in bursts: is it distance or pairing that counts?
  mulK, addK  v_mul_f32 / v_add_f32 chains
  trans*      v_rcp_f32 / v_rsq_f32 between FMAs: what a transcendental costs and what it overlaps with
  cmp*        v_cmp + v_cndmask, v_cmp + s_or + never-taken branch (the range guards), s_and_saveexec regions
  mov*        v_mov_b32 rotation cost
"""
import sys

N_BODY = 96          # instructions per loop body (patterns are padded / repeated to this length)
ITERS = 6000

def chains(k, op="fma"):
    out = []
    for i in range(N_BODY):
        r = 10 + (i % k)
        if op == "fma":
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        elif op == "lit":
            out.append(f"v_fmac_f32 v{r}, 0x3a83126f, v{r}")
        elif op == "mul":
            out.append(f"v_mul_f32 v{r}, v{r}, v8")
        elif op == "add":
            out.append(f"v_add_f32 v{r}, v{r}, v9")
        elif op == "fmamk":
            out.append(f"v_fmamk_f32 v{r}, v{r}, 0x3f7fbe77, v9")
    return out

def bursts(b, k):
    """k chains, each issued in bursts of b dependent instructions: A A B B ..."""
    out = []
    i = 0
    while len(out) < N_BODY:
        r = 10 + (i % k)
        for _ in range(b):
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    return out[:N_BODY]

def stage_shaped():
    """three accumulator chains (x, y, z) each reading ANOTHER value, as the Cash-Karp stages do"""
    out = []
    for i in range(N_BODY // 3):
        for c in range(3):
            out.append(f"v_fmac_f32 v{10 + c}, 0x3a83126f, v{20 + (i % 5) * 3 + c}")
    return out

def trans(every, k, top="v_rcp_f32", dep=False):
    """one transcendental every `every` instructions among k fma chains; dep: the next fma uses its result"""
    out = []
    i = 0
    while len(out) < N_BODY:
        if i % every == 0:
            out.append(f"{top} v40, v41")
            if dep:
                out.append("s_nop 0")
                out.append("v_fma_f32 v42, v40, v8, v9")
        else:
            r = 10 + (i % k)
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    return out[:N_BODY]

def cmp_sel(k):
    out = []
    i = 0
    while len(out) < N_BODY:
        r = 10 + (i % k)
        if i % 8 == 7:
            out.append(f"v_cmp_lt_f32 vcc, v{r}, v9")
            out.append("s_nop 1")
            out.append(f"v_cndmask_b32 v43, v8, v9, vcc")
        else:
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    return out[:N_BODY]

def guard(every, k):
    """the range guard: two compares, s_or, a never-taken branch, every `every` instructions"""
    out = []
    i = 0
    lab = 0
    while len(out) < N_BODY:
        r = 10 + (i % k)
        if i % every == every - 1:
            out.append(f"v_cmp_nle_f32 vcc, s30, v{r}")
            out.append(f"v_cmp_nge_f32 s[42:43], s31, v{r}")
            out.append("s_or_b64 vcc, vcc, s[42:43]")
            out.append(f"s_cbranch_vccnz 9f")
        else:
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    return out[:N_BODY]

def guard_deferred(every, k):
    """the same compares, their masks OR-ed into an SGPR pair, no branch (one test per body at the end)"""
    out = []
    i = 0
    while len(out) < N_BODY - 1:
        r = 10 + (i % k)
        if i % every == every - 1:
            out.append(f"v_cmp_nle_f32 vcc, s30, v{r}")
            out.append(f"v_cmp_nge_f32 s[42:43], s31, v{r}")
            out.append("s_or_b64 s[34:35], s[34:35], vcc")
            out.append("s_or_b64 s[34:35], s[34:35], s[42:43]")
        else:
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    out = out[:N_BODY - 2]
    out.append("s_cmp_lg_u64 s[34:35], 0")
    out.append("s_cbranch_scc1 9f")
    return out

def guard_class(every, k):
    """v_cmp_class-free variant: ONE compare per guard via a biased unsigned compare of the bit pattern"""
    out = []
    i = 0
    while len(out) < N_BODY - 2:
        r = 10 + (i % k)
        if i % every == every - 1:
            out.append(f"v_subrev_u32 v44, s36, v{r}")
            out.append(f"v_cmp_lt_u32 vcc, s37, v44")
            out.append("s_or_b64 s[34:35], s[34:35], vcc")
        else:
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    out = out[:N_BODY - 2]
    out.append("s_cmp_lg_u64 s[34:35], 0")
    out.append("s_cbranch_scc1 9f")
    return out

def saveexec(every, k):
    out = []
    i = 0
    while len(out) < N_BODY:
        r = 10 + (i % k)
        if i % every == every - 1:
            out.append(f"v_cmp_lt_f32 vcc, v9, v{r}")     # true for all lanes (values near 1 > 0.001)
            out.append("s_and_saveexec_b64 s[42:43], vcc")
            out.append("s_cbranch_execz 9f")
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
            out.append("s_or_b64 exec, exec, s[42:43]")
        else:
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    return out[:N_BODY]

def movs(k, nm):
    """k fma chains with nm v_mov per 16 instructions (the state rotation of the step loop)"""
    out = []
    i = 0
    while len(out) < N_BODY:
        if i % 16 < nm:
            out.append(f"v_mov_b32 v{45 + (i % 16)}, v{10 + (i % k)}")
        else:
            r = 10 + (i % k)
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    return out[:N_BODY]

def mixed_indep_pairs():
    """pairs (fma, mul) of different chains - does the opcode mix matter?"""
    out = []
    for i in range(N_BODY // 2):
        a = 10 + (i % 4)
        out.append(f"v_fma_f32 v{a}, v{a}, v8, v9")
        out.append(f"v_mul_f32 v{14 + (i % 4)}, v{14 + (i % 4)}, v8")
    return out

def sgpr_operand(k):
    out = []
    for i in range(N_BODY):
        r = 10 + (i % k)
        out.append(f"v_fma_f32 v{r}, v{r}, s38, v9")
    return out

PATTERNS = {}
for k in (1, 2, 3, 4, 5, 6, 8):
    PATTERNS[f"d{k}"] = chains(k)
for k in (1, 2, 3, 4, 6):
    PATTERNS[f"lit{k}"] = chains(k, "lit")
for k in (1, 2, 3, 4):
    PATTERNS[f"mul{k}"] = chains(k, "mul")
PATTERNS["add1"] = chains(1, "add")
PATTERNS["add3"] = chains(3, "add")
PATTERNS["fmamk3"] = chains(3, "fmamk")
PATTERNS["aabb2"] = bursts(2, 2)
PATTERNS["aabb3"] = bursts(2, 3)
PATTERNS["aabb4"] = bursts(2, 4)
PATTERNS["aaabbb3"] = bursts(3, 3)
PATTERNS["stage"] = stage_shaped()
PATTERNS["sgpr3"] = sgpr_operand(3)
PATTERNS["rcp8_k3"] = trans(8, 3)
PATTERNS["rcp8_k3_dep"] = trans(8, 3, dep=True)
PATTERNS["rcp4_k3"] = trans(4, 3)
PATTERNS["rcp2_k4"] = trans(2, 4)
PATTERNS["rsq8_k3"] = trans(8, 3, "v_rsq_f32")
PATTERNS["sqrt8_k3"] = trans(8, 3, "v_sqrt_f32")
PATTERNS["rcp8_k1"] = trans(8, 1)
PATTERNS["cmpsel_k3"] = cmp_sel(3)
PATTERNS["guard16_k3"] = guard(16, 3)
PATTERNS["guard32_k3"] = guard(32, 3)
PATTERNS["guarddef16_k3"] = guard_deferred(16, 3)
PATTERNS["guardcls16_k3"] = guard_class(16, 3)
PATTERNS["saveexec16_k3"] = saveexec(16, 3)
PATTERNS["saveexec32_k3"] = saveexec(32, 3)
PATTERNS["mov0_k3"] = movs(3, 0)
PATTERNS["mov2_k3"] = movs(3, 2)
PATTERNS["mov4_k3"] = movs(3, 4)
PATTERNS["mixpairs"] = mixed_indep_pairs()


def mix(xs, period, k=3):
    """fma chains with the instruction(s) xs inserted once per `period` slots"""
    out = []
    i = 0
    while len(out) < N_BODY:
        if i % period == period - 1:
            out.extend(xs)
        else:
            r = 10 + (i % k)
            out.append(f"v_fma_f32 v{r}, v{r}, v8, v9")
        i += 1
    return out[:N_BODY]

def allx(xs):
    out = []
    while len(out) < N_BODY:
        out.extend(xs)
    return out[:N_BODY]

SET2 = {}
SET2["base_k3"] = chains(3)
X = {
  "mul_vs":    ["v_mul_f32 v50, s38, v20"],
  "mul_vv":    ["v_mul_f32 v50, v8, v20"],
  "add_vs":    ["v_add_f32 v50, s38, v20"],
  "subrev_vs": ["v_subrev_f32 v50, s38, v20"],
  "fma_svv":   ["v_fma_f32 v50, s38, v20, v21"],
  "fma_vsv":   ["v_fma_f32 v50, v20, s38, v21"],
  "fma_vvv":   ["v_fma_f32 v50, v20, v22, v21"],
  "fma_neg":   ["v_fma_f32 v50, -v20, v22, 1.0"],
  "mul_e64neg":["v_mul_f32_e64 v50, v20, -v22"],
  "fmac_lit":  ["v_fmac_f32 v50, 0x3a83126f, v20"],
  "fmac_vs":   ["v_fmac_f32 v50, s38, v20"],
  "cmp_vcc_vv":["v_cmp_lt_f32 vcc, v20, v21"],
  "cmp_vcc_sv":["v_cmp_lt_f32 vcc, s38, v21"],
  "cmp_e64_vv":["v_cmp_lt_f32 s[42:43], v20, v21"],
  "cmp_e64_sv":["v_cmp_lt_f32 s[42:43], s38, v21"],
  "cmp_abs":   ["v_cmp_lt_f32 s[42:43], |v20|, |v21|"],
  "cmp_i32_sv":["v_cmp_gt_i32 vcc, s40, v21"],
  "cndmask_vcc":["v_cndmask_b32 v50, v20, v21, vcc"],
  "cndmask_e64":["v_cndmask_b32 v50, v20, v21, s[44:45]"],
  "mov_vv":    ["v_mov_b32 v50, v20"],
  "mov_vs":    ["v_mov_b32 v50, s38"],
  "mov_lit":   ["v_mov_b32 v50, 0x3a83126f"],
  "add_u32":   ["v_add_u32 v50, 1, v50"],
  "and_or":    ["v_and_or_b32 v50, v20, s38, 1.0"],
  "s_nop0":    ["s_nop 0"],
  "s_nop1":    ["s_nop 1"],
  "s_or":      ["s_or_b64 s[42:43], s[44:45], s[46:47]"],
  "s_add":     ["s_add_i32 s48, s48, 1"],
  "s_cmp_sel": ["s_cmp_eq_u32 s48, 0", "s_cselect_b64 s[42:43], -1, 0"],
  "s_mov":     ["s_mov_b32 s48, 0x1000000"],
  "saveexec":  ["s_and_saveexec_b64 s[42:43], s[44:45]", "s_or_b64 exec, exec, s[42:43]"],
  "branch_nt": ["s_cbranch_scc0 9f"],
  "branch_vccz_nt": ["s_cbranch_vccnz 9f"],
  "rcp":       ["v_rcp_f32 v50, v41"],
  "rsq":       ["v_rsq_f32 v50, v41"],
  "rcp_nop_fma":["v_rcp_f32 v50, v41", "s_nop 0", "v_fma_f32 v51, -v41, v50, 1.0"],
  "pk_fma":    ["v_pk_fma_f32 v[50:51], v[20:21], v[22:23], v[24:25]"],
  "pk_mul":    ["v_pk_mul_f32 v[50:51], v[20:21], v[22:23]"],
  "dpp_mov":   ["v_mov_b32_dpp v50, v20 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf"],
  "max_abs":   ["v_max_f32 v50, |v20|, |v21|"],
  "max3":      ["v_max3_f32 v50, |v20|, |v21|, |v22|"],
  "med3":      ["v_med3_f32 v50, v20, v21, v22"],
  "ldexp":     ["v_ldexp_f32 v50, v20, v21"],
  "cvt":       ["v_cvt_f32_i32 v50, v20"],
  "lshr":      ["v_lshrrev_b32 v50, 23, v20"],
}
for n, xs in X.items():
    SET2[f"{n}_p4"] = mix(xs, 4)
for n in ("mul_vs", "fma_svv", "cmp_vcc_sv", "cmp_e64_vv", "mov_vv", "s_or", "s_nop0", "rcp", "fmac_vs", "add_vs"):
    SET2[f"{n}_p2"] = mix(X[n], 2)
for n in ("mul_vs", "fma_svv", "fma_vvv", "mul_vv", "cmp_vcc_vv", "cmp_vcc_sv", "mov_vv", "cndmask_vcc", "pk_fma", "max3", "s_or"):
    SET2[f"{n}_all"] = allx(X[n])
if len(sys.argv) > 2 and sys.argv[2] == "2":
    PATTERNS = SET2

# ---- the dense RK kernel's real per-step tail (blocks 92 / 93 / 94 / 95 / 96 of trace_kernel<1, false, false, true, 0>, round 6 ISA), verbatim but for
# register numbers and the two branches: what its 30 SGPR-touching vector instructions cost, and what they would cost from VGPRs or spaced apart
import re
TAIL = """
v_fmac_f32_e32 v43, v53, v36
v_fmac_f32_e32 v42, v52, v36
v_subrev_f32_e32 v44, s24, v43
v_fmac_f32_e32 v41, v51, v36
v_subrev_f32_e32 v45, s25, v42
v_mul_f32_e32 v4, v44, v44
v_subrev_f32_e32 v46, s99, v41
v_fmac_f32_e32 v4, v45, v45
v_fmac_f32_e32 v4, v46, v46
v_cmp_nle_f32_e32 vcc, s92, v4
v_cmp_nge_f32_e64 s[0:1], s26, v4
s_or_b64 vcc, vcc, s[0:1]
s_cbranch_vccnz 9f
v_rsq_f32_e32 v7, v4
s_nop 0
v_mul_f32_e32 v47, v4, v7
v_fma_f32 v4, -v47, v47, v4
v_mul_f32_e32 v7, 0.5, v7
v_fmac_f32_e32 v47, v4, v7
v_mul_f32_e32 v36, v36, v6
v_mul_f32_e32 v9, 0x3f866666, v36
v_add_f32_e32 v9, 0x3d4ccccd, v9
v_sub_f32_e32 v6, s24, v37
v_add_f32_e32 v10, 1.0, v9
v_add_f32_e32 v9, s95, v9
v_sub_f32_e32 v4, s25, v38
v_mul_f32_e32 v8, s2, v6
v_cmp_le_f32_e64 s[0:1], v60, v9
v_mul_f32_e32 v9, 0x3f8147ae, v36
v_sub_f32_e32 v7, s99, v39
v_fma_f32 v6, s6, v4, v8
v_mul_f32_e32 v9, s36, v9
v_fmac_f32_e32 v6, s7, v7
v_add_f32_e32 v9, v22, v9
v_cmp_le_f32_e64 s[8:9], |v6|, v9
v_cmp_le_f32_e32 vcc, v60, v10
s_and_b64 s[22:23], s[0:1], s[8:9]
s_or_b64 s[8:9], vcc, s[22:23]
v_cmp_lt_f32_e64 s[0:1], s58, v47
v_cmp_lt_f32_e64 s[4:5], v47, v40
s_nor_b64 s[8:9], s[8:9], s[0:1]
s_and_saveexec_b64 s[14:15], s[8:9]
s_xor_b64 s[8:9], exec, s[14:15]
v_add_u32_e32 v34, 1, v34
s_or_saveexec_b64 s[14:15], s[8:9]
v_mul_f32_e32 v48, v0, v3
v_mul_f32_e32 v49, v1, v3
v_mul_f32_e32 v50, v2, v3
v_cndmask_b32_e64 v40, v40, v47, s[4:5]
v_mov_b32_e32 v6, 1
v_mov_b32_e32 v1, v41
v_mov_b32_e32 v2, v42
v_mov_b32_e32 v3, v43
v_mov_b32_e32 v56, v50
v_mov_b32_e32 v55, v49
v_mov_b32_e32 v54, v48
v_mov_b32_e32 v60, v47
s_xor_b64 exec, exec, s[14:15]
s_or_b64 exec, exec, s[14:15]
""".strip().split("\n")
SMAP = {"s24": "s50", "s25": "s51", "s99": "s52", "s92": "s53", "s26": "s54", "s95": "s55", "s2": "s56", "s6": "s57", "s7": "s58", "s36": "s59", "s58": "s60",
        "s[0:1]": "s[62:63]", "s[4:5]": "s[64:65]", "s[8:9]": "s[66:67]", "s[14:15]": "s[68:69]", "s[22:23]": "s[70:71]"}
VMAP_OF_S = {"s24": "v69", "s25": "v70", "s99": "v71", "s92": "v72", "s26": "v73", "s95": "v74", "s2": "v75", "s6": "v76", "s7": "v77", "s36": "v78", "s58": "v79"}
def remap(line, s_to_v=False):
    def vr(m): return "v%d" % (int(m.group(1)) + 8)
    line = re.sub(r"\bv(\d+)\b", vr, line)
    def sr(m):
        t = m.group(0)
        if s_to_v and t in VMAP_OF_S: return VMAP_OF_S[t]
        return SMAP.get(t, t)
    return re.sub(r"s\[\d+:\d+\]|\bs\d+\b", sr, line)
def tail(s_to_v=False, order=None, rep=1):
    ls = TAIL if order is None else [TAIL[i] for i in order]
    return [remap(l, s_to_v) for l in ls] * rep
# a hand-spaced order: every instruction that reads or writes an SGPR separated from the next such one by a plain one where the dependencies allow
SPACED = [0, 2, 1, 4, 3, 6, 5, 7, 8,   9, 19, 10, 20, 11, 12,  13, 14, 15, 16, 17, 18,
          22, 21, 25, 23, 24, 29, 26, 28, 30, 45, 31, 46, 32, 47, 33, 27, 50, 34, 51, 35, 52, 36, 37, 38, 39, 40, 41, 42, 43, 44, 48, 49, 53, 54, 55, 56, 57, 58]
assert sorted(SPACED) == list(range(len(TAIL))), (len(TAIL), sorted(set(range(len(TAIL))) - set(SPACED)), [x for x in SPACED if SPACED.count(x) > 1])
SET3 = {"base_k3": chains(3), "tail_real": tail(rep=2), "tail_vgpr": tail(True, rep=2), "tail_spaced": tail(order=SPACED, rep=2), "tail_spaced_vgpr": tail(True, order=SPACED, rep=2)}
if len(sys.argv) > 2 and sys.argv[2] == "3":
    PATTERNS = SET3

HEADER = r'''// GENERATED by gen.py - do not edit.  hipcc --offload-arch=gfx950 -O2 issue_rules.hip -o issue_rules
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
'''

def kernel(name, body):
    nvalu = sum(1 for l in body if l.startswith("v_"))
    lines = ["v_mov_b32 v8, 0x3f7fbe77", "v_mov_b32 v9, 0x3a83126f", "v_mov_b32 v41, 0x3fc00000"]
    for r in range(10, 80):
        if r in (41,):
            continue
        lines.append(f"v_mov_b32 v{r}, 1.0")
    lines += ["s_mov_b32 s30, 0x00800000", "s_mov_b32 s31, 0x7e800000", "s_mov_b64 s[34:35], 0",
              "s_mov_b64 s[44:45], -1", "s_mov_b64 vcc, 0", "s_mov_b64 s[46:47], 0", "s_mov_b32 s48, 5", "s_mov_b32 s36, 0x00800000", "s_mov_b32 s37, 0x7e000000", "s_mov_b32 s38, 0x3f7fbe77",
              "s_mov_b32 s50, 0", "s_mov_b32 s51, 0", "s_mov_b32 s52, 0", "s_mov_b32 s53, 0", "s_mov_b32 s54, 0x7e800000", "s_mov_b32 s55, 1.0", "s_mov_b32 s56, 0.5", "s_mov_b32 s57, 0.5", "s_mov_b32 s58, 0.5", "s_mov_b32 s59, 1.0", "s_mov_b32 s60, 0x41a00000", "v_mov_b32 v72, 0", "v_mov_b32 v73, 0x7e800000", "v_mov_b32 v79, 0x41a00000", "v_mov_b32 v44, 0x3f7fbe77", "v_mov_b32 v69, 0", "v_mov_b32 v70, 0", "v_mov_b32 v71, 0", "s_mov_b32 s40, %1", "1:"]
    lines += body
    lines += ["s_sub_u32 s40, s40, 1", "s_cmp_lg_u32 s40, 0", "s_cbranch_scc1 1b", "9:",
              "v_add_f32 %0, v10, v11", "v_add_f32 %0, %0, v12", "v_add_f32 %0, %0, v13", "v_add_f32 %0, %0, v40",
              "v_add_f32 %0, %0, v42", "v_add_f32 %0, %0, v43"]
    asm = "\n".join(f'        "{l}\\n"' for l in lines)
    clob = ", ".join(f'"v{r}"' for r in range(8, 80)) + ', "s30", "s31", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s34", "s35", "s36", "s37", "s38", "s40", "vcc", "scc", "memory"'
    return nvalu, f'''
__global__ __launch_bounds__(256) void k_{name}(float* out, int iters) {{
    float r;
    asm volatile(
{asm}
        : "=v"(r) : "s"(iters) : {clob});
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}}
'''

def main():
    out = [HEADER]
    table = []
    for name, body in PATTERNS.items():
        pass
        nvalu, src = kernel(name, body)
        out.append(src)
        table.append((name, len(body), nvalu))
    out.append("struct Pat { const char* name; void (*fn)(float*, int); int n, nvalu; };\nstatic Pat pats[] = {\n")
    for name, n, nv in table:
        out.append(f'    {{"{name}", k_{name}, {n}, {nv}}},\n')
    out.append("};\n")
    out.append(r'''
int main(int argc, char** argv) {
    const int iters = %d;
    float* out; CK(hipMalloc(&out, (size_t)256 * 8 * 256 * sizeof(float)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int waves[] = {1, 2, 4, 6, 8};
    printf("%%-16s %%5s %%5s |", "pattern", "n", "valu");
    for (int w : waves) printf(" %%7dw", w);
    printf("   (cycles per instruction per SIMD at 2.4 GHz, all instructions of the body counted)\n");
    for (const Pat& p : pats) {
        if (argc > 1 && !strstr(p.name, argv[1])) continue;
        printf("%%-16s %%5d %%5d |", p.name, p.n, p.nvalu);
        for (int w : waves) {
            const int blocks = 256 * w;       // 256 threads = 4 waves = one per SIMD of a CU; w blocks per CU
            hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, out, 200);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(p.fn, dim3(blocks), dim3(256), 0, 0, out, iters);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double instr_per_simd = (double)iters * p.n * w;
            printf(" %%8.2f", best * 1e-3 * 2.4e9 / instr_per_simd);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
''' % ITERS)
    open(sys.argv[1] if len(sys.argv) > 1 else "issue_rules.hip", "w").write("".join(out))

if __name__ == "__main__":
    main()
