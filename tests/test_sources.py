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


def test_default_build_instantiates_no_fused_ladder_and_no_experiment_macros():
    src = open(os.path.join(ROOT, "bhusie_amd", "csrc", "bhray_kernels.hip")).read()
    assert "BHRAY_EXP_" not in src and "#define BHRAY_WITH_FUSED 0" in src
    # the fused ladder's code lives in its own include, reached only under BHRAY_WITH_FUSED
    for m in re.finditer(r'#include "bhray_fused.inc"', src):
        before = src[:m.start()]
        assert before.rfind("#if BHRAY_WITH_FUSED") > before.rfind("#endif"), "bhray_fused.inc included outside an #if BHRAY_WITH_FUSED block"
    mk = open(os.path.join(ROOT, "bhusie_amd", "csrc", "Makefile")).read()
    assert "-DBHRAY_WITH_FUSED=1" in mk and "fused:" in mk


def test_default_build_has_no_pair_march_and_the_pair_step_is_next_ray_rk_on_2_vectors():
    """bhray_pair.inc is a build option (make pair).  Its integrator step must be next_ray_rk line for line with the scalar helpers
    replaced by their 2-vector counterparts - the GPU tests compare frames byte for byte, this catches a drift between the two texts
    at review time."""
    src = open(os.path.join(ROOT, "bhusie_amd", "csrc", "bhray_kernels.hip")).read()
    assert "#define BHRAY_WITH_PAIR 0" in src
    inc = open(os.path.join(ROOT, "bhusie_amd", "csrc", "bhray_pair.inc")).read()
    assert inc.lstrip().startswith("//") and "#if BHRAY_WITH_PAIR" in inc and inc.rstrip().endswith("#endif  // BHRAY_WITH_PAIR")
    code = inc[inc.index("#if BHRAY_WITH_PAIR"):]
    assert "__global__" in code                                   # the kernel lives entirely inside the #if

    def body(text, name):
        i = text.index(name)
        i = text.index("{", i)
        depth, j = 0, i
        while True:
            depth += {"{": 1, "}": -1}.get(text[j], 0)
            if depth == 0:
                return text[i + 1:j]
            j += 1

    def stage_lines(b):
        out = []
        for line in b.splitlines():
            line = line.split("//")[0].strip()
            if re.match(r"const (F3|P3) (K[1-6]|e|ds|cr) =", line):
                out.append(line)
        return out

    scalar = stage_lines(body(src, "void next_ray_rk(F3 q0"))
    packed = stage_lines(body(inc, "void next_ray_rk_pair(P3T<V> q0"))
    assert len(scalar) == len(packed) == 9
    for a, b in zip(scalar, packed):
        b = b.replace("P3", "F3").replace("pmadd3", "fmadd3").replace("pcross", "fcross")
        b = re.sub(r"sp<V>\(([A-Z0-9]+)\)", r"\1", b)
        assert a == b, (a, b)
    mk = open(os.path.join(ROOT, "bhusie_amd", "csrc", "Makefile")).read()
    assert "-DBHRAY_WITH_PAIR=1" in mk and "pair:" in mk
