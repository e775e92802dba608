"""CPU: properties of the kernel SOURCES that no GPU test can see.

bhray_step.inc holds the integrator step of the trace kernel twice - the dense form and the lean form (one exec-mask region per step,
for a wave that runs alone) - because three attempts at one parametrised text changed the dense build's code generation.  The two
forms must be the same operations on the same values; only the control flow around them may differ.  This test diffs them: outside
an explicit list of control-flow lines (how a lane leaves the march and where `it` is counted) they must be identical, token for token."""
import difflib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the ONLY lines that may differ: each form's way of (1) entering the step / ending a ray that has used up its iterations,
# (2) counting the step (`it`), (3) leaving on a disk hit
LEAN_ONLY = ["{", "const bool go = (mode == M_REL) & (it < H.max_iter);", "if ((mode == M_REL) & !go) mode = M_FINISH;", "if (go) {",
             "it++;", "it--;", "} else {", "if (amount < 0.005f) { mode = M_FINISH; it--; }", "}"]
DENSE_ONLY = ["if (mode == M_REL) {", "if (it >= H.max_iter) {", "mode = M_FINISH;", "} else {", "continue;", "}",
              "if (amount < 0.005f) mode = M_FINISH; else it++;", "} else {", "it++;"]


def _forms():
    t = open(os.path.join(ROOT, "bhusie_amd", "csrc", "bhray_step.inc")).read()
    lean = t[t.index("#if BHRAY_STEP_LEAN"):t.index("#else")]
    dense = t[t.index("#else"):t.index("#endif", t.index("#else"))]

    def stmts(txt):
        out = []
        for line in txt.splitlines()[1:]:
            line = re.sub(r"//.*", "", line).strip()
            if line:
                out.append(re.sub(r"\s+", " ", line))
        return out
    return stmts(lean), stmts(dense)


def test_the_two_forms_of_the_integrator_step_are_the_same_operations():
    lean, dense = _forms()
    assert len(lean) > 60 and len(dense) > 60
    removed, added = [], []
    for d in difflib.unified_diff(lean, dense, lineterm="", n=0):
        if d.startswith(("---", "+++", "@@")):
            continue
        (removed if d[0] == "-" else added).append(d[1:])
    assert removed == LEAN_ONLY, removed
    assert added == DENSE_ONLY, added
    # and none of the lines that differ computes anything but the iteration count / the mode
    for line in removed + added:
        assert not re.search(r"\b(cpos|cdir|ppos|pdir|rkpos|rkdir|rkh|qrel|dist_c|cpos_dist|closest|cold|next_ray|hit_black_hole|black_hole_culls)\b", line), line


def test_the_product_tree_holds_no_measured_and_rejected_variants():
    """the fused ladder (BHRAY_F_FUSED), the pair march and the experiment flags of rounds 3-5 are patches under
    profiles/variants_src/, not code the product builds; the trace kernel has at most 32 instantiations."""
    csrc = os.path.join(ROOT, "bhusie_amd", "csrc")
    src = open(os.path.join(csrc, "bhray_kernels.hip")).read()
    for gone in ("BHRAY_EXP_", "BHRAY_EXPERIMENT_", "BHRAY_WITH_FUSED", "BHRAY_WITH_PAIR", "BHRAY_BVH_PREFETCH", "bhray_fused.inc", "bhray_pair.inc"):
        assert not re.search(r"^\s*#\s*(if|ifdef|ifndef|define|include).*" + re.escape(gone), src, re.M), gone
    assert not os.path.exists(os.path.join(csrc, "bhray_fused.inc")) and not os.path.exists(os.path.join(csrc, "bhray_pair.inc"))
    assert "BHRAY_F_FUSED " not in open(os.path.join(ROOT, "include", "bhray.h")).read().replace("(1u << 6 was BHRAY_F_FUSED,", "")
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "fused:" not in mk and "pair:" not in mk and "stack2:" in mk
    vs = os.path.join(ROOT, "profiles", "variants_src")
    assert os.path.exists(os.path.join(vs, "r06_fused_ladder_and_pair_march.patch")) and os.path.exists(os.path.join(vs, "README.md"))
    # instantiations: every explicit trace_kernel<...> the launchers can reach
    combos = set()
    for meth in (0, 1):
        for models in (False, True):
            for dense in (False, True):
                for count in (False, True):
                    for ev in (0, 1, 2):
                        if _trace_variant_exists(src, ev, models, dense, count):
                            combos.add((meth, models, dense, count, ev))
    assert 16 <= len(combos) <= 32, len(combos)


def _trace_variant_exists(src, ev, models, dense, count):
    """mirrors trace_variant_exists() of bhray_kernels.hip: which (evaluation, mesh, dense, counting) builds are instantiated"""
    m = re.search(r"constexpr bool trace_variant_exists\(int eval, bool models, bool dense, bool count\) \{\s*return (.*?);\s*\}", src, re.S)
    assert m, "trace_variant_exists() not found in bhray_kernels.hip"
    expr = m.group(1).replace("&&", " and ").replace("||", " or ").replace("!", " not ").replace("not =", "!=")
    return bool(eval(expr, {}, {"eval": ev, "models": models, "dense": dense, "count": count}))
