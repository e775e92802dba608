import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import bhusie_amd as B
from tests import common as T
tex = T.textures()
u = T.uniforms(integration_method=1)
cfg = B.ladder_from_base((24, 14), 3, 3)
one = B.RayPass(cfg, device=0); one.set_textures(*tex); one.set_uniforms(*u); one.render(); want = one.read_hdr()
print("plain ok", flush=True)
for spec in (0, 2):
    fz = B.RayPass(cfg, device=0, fused=True, speculative_levels=spec, frames_in_flight=1); fz.set_textures(*tex); fz.set_uniforms(*u)
    print("created", spec, flush=True)
    fz.render()
    print("enqueued", flush=True)
    fz.sync()
    print("synced", flush=True)
    got = fz.read_hdr()
    print("spec", spec, "pixels differing:", int((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).sum()), flush=True)
    fz.close()
