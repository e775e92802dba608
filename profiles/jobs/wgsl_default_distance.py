"""GPU job: the DEFAULT (contract) kernels against the frames made by executing the reference's shader text (tests/golden/wgsl_exec.npz).
Prints one JSON line per scene: class differences, pixels beyond 1e-4 per channel / against the pixel's norm, medians."""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bhusie_amd as B

g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "wgsl_exec.npz"))
tex = (g["t_temp"], g["t_disk"], g["t_sky"])
names = sorted(k[:-6] for k in g.files if k.endswith(".sizes"))
tot = dict(px=0, cls=0, ch=0, nrm=0)
for name in names:
    u = tuple(g[f"{name}.{k}"].tobytes() for k in ("camera", "black_hole", "details"))
    sizes = [tuple(int(v) for v in s) for s in g[f"{name}.sizes"]]
    rp = B.RayPass(B.ladder_from_base(sizes[0], 3, len(sizes)), device=0)
    rp.set_textures(*tex)
    if int(g[f"{name}.mesh"][0]):
        with tempfile.NamedTemporaryFile(suffix=".obj", delete=False) as f:
            f.write(g["mesh.obj"].tobytes())
        rp.upload_model(B.load_model(f.name))
    rp.set_uniforms(*u)
    rp.render()
    for l in range(len(sizes)):
        got, want = rp.read_level(l), g[f"{name}.level{l}"]
        ok = ~(np.isnan(want).any(-1) | np.isnan(got).any(-1))
        nan_same = bool(np.array_equal(np.isnan(want).any(-1), np.isnan(got).any(-1)))
        cls = int((got[..., 3][ok] != want[..., 3][ok]).sum())
        same = ok & (got[..., 3] == want[..., 3])
        a, b = got[same][:, :3], want[same][:, :3]
        rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
        nrm = np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-3)
        rec = dict(scene=name, level=l, pixels=int(ok.sum()), nan_same=nan_same, class_differences=cls, beyond_1e4_per_channel=int((rel.max(-1) > 1e-4).sum()),
                   beyond_1e4_of_norm=int((nrm > 1e-4).sum()), median_of_norm=float(np.median(nrm)) if len(nrm) else 0.0, max_of_norm=float(nrm.max(initial=0.0)))
        print(json.dumps(rec))
        tot["px"] += rec["pixels"]; tot["cls"] += cls; tot["ch"] += rec["beyond_1e4_per_channel"]; tot["nrm"] += rec["beyond_1e4_of_norm"]
print(json.dumps(dict(total=tot)))
