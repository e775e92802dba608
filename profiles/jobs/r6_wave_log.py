"""How full is the device?  A measurement build (-DBHRAY_WAVE_LOG=1: every trace wave leaves a record - start, end on the 100 MHz clock, HW_ID, integrator
steps issued) renders blocks of frames the way bench.py does (22 frame slots, speculative levels 2) and this script reconstructs, from the records, how many
trace waves were resident over time.  usage: BHRAY_LIB=profiles/variants/libbhray_wlog.so python profiles/jobs/r6_wave_log.py [steps] [integrator] [frames_in_flight]"""
import os, sys, time, argparse, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "64")
import bhusie_amd as B
import bench
from bhusie_amd import renderer as R
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
integ = sys.argv[2] if len(sys.argv) > 2 else "rk"
fif = int(sys.argv[3]) if len(sys.argv) > 3 else 22
a = argparse.Namespace(workload="disk", integrator=integ, max_iterations=2000, bvh="reference")
tex, cam, bh, det, model = bench.build_scene(a)
cfg = B.ladder_for_frame((1920, 1080), 3, 4)
rp = B.RayPass(cfg, device=0, frames_in_flight=fif, speculative_levels=2, timing="sparse")
rp.set_textures(*tex); rp.set_uniforms(cam.uniform(), bh.uniform(), det.uniform())
lib = R.lib()
CAP = 1 << 20
buf = C.c_void_p()
assert lib.bhray_host_alloc(CAP * 32, C.byref(buf)) == 0
lib.bhray_debug_wave_log.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint)]
arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint64)), shape=(CAP, 4))
def block(n):
    rp.sync(); t0 = time.perf_counter()
    for _ in range(n): rp.render()
    rp.sync(); return time.perf_counter() - t0
for _ in range(3): block(max(steps, 32))
for rep in range(3):
    assert lib.bhray_debug_wave_log(buf, CAP, None) == 0
    el = block(steps)
    n = C.c_uint(0)
    assert lib.bhray_debug_wave_log(None, 0, C.byref(n)) == 0
    k = min(n.value, CAP)
    rec = arr[:k].copy()
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    base = t0.min(); span = (t1.max() - base) / 100.0          # us
    life = (t1 - t0) / 100.0
    # resident waves over time: +1 at start, -1 at end
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    lvl = np.cumsum(ev[:, 1]); dt = np.diff(ev[:, 0], append=ev[-1, 0])
    avg = float((lvl * dt).sum() / max(1, (t1.max() - base)))
    hist = np.zeros(8)
    for lo in range(8):
        m = (lvl / 1024.0 >= lo) & (lvl / 1024.0 < lo + 1)
        hist[lo] = dt[m].sum() / max(1, (t1.max() - base))
    ws = (rec[:, 3] & 0xffffff).astype(np.int64)
    # launches: waves that share a queue address and start within the same episode (a queue is reused by the slot's next frame: split at gaps > the launch's life)
    qid = (rec[:, 3] >> 32).astype(np.int64)
    launches = []
    for q in np.unique(qid):
        m = np.where(qid == q)[0]; o = m[np.argsort(t0[m])]
        cut = np.where(np.diff(t0[o]) > 20000)[0]          # 200 us without a new wave: the next frame's launch on this queue
        for seg in np.split(o, cut + 1):
            launches.append((t0[seg].min(), t0[seg].max(), t1[seg].max(), len(seg), np.percentile(t0[seg], 50), np.percentile(t1[seg], 50), np.percentile(t1[seg], 90)))
    L = np.array(launches, dtype=np.float64)
    print(f"   {len(L)} launches: waves per launch mean {L[:,3].mean():.0f}; first->last wave start (dispatch spread) mean {(L[:,1]-L[:,0]).mean()/100:.1f} us, median {np.median(L[:,1]-L[:,0])/100:.1f}, p90 {np.percentile(L[:,1]-L[:,0],90)/100:.1f};"
          f" launch life (first start -> last end) mean {(L[:,2]-L[:,0]).mean()/100:.1f} us; median wave end at {np.mean(L[:,5]-L[:,0])/100:.1f} us, p90 wave end {np.mean(L[:,6]-L[:,0])/100:.1f} us after the first start")
    evl = np.concatenate([np.stack([L[:,0], np.ones(len(L))], 1), np.stack([L[:,2], -np.ones(len(L))], 1)]); evl = evl[np.argsort(evl[:,0], kind="stable")]
    lv = np.cumsum(evl[:,1]); dtl = np.diff(evl[:,0], append=evl[-1,0])
    print(f"   launches alive (first wave start .. last wave end) over the span: mean {(lv*dtl).sum()/max(1,(t1.max()-base)):.2f}, max {lv.max():.0f}")
    print(f"block of {steps} frames ({integ}, {fif} slots): host {el*1e3:.3f} ms = {el/steps*1e3:.4f} ms/frame; {k} trace waves, device span {span/1e3:.3f} ms; wave life mean {life.mean():.1f} us, median {np.median(life):.1f}, max {life.max():.1f}; "
          f"resident trace waves per SIMD: mean {avg/1024:.2f}; share of the span with [0,1) [1,2) .. waves per SIMD: {np.round(hist, 3).tolist()}; wave-steps per wave mean {ws.mean():.0f}; sum of wave life {life.sum()/1e3:.1f} ms = {life.sum()/span/1024:.2f} per SIMD")
    # per SIMD: how evenly? HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13]; XCC_ID[3:0]
    hw = rec[:, 2]; simd = ((hw >> 4) & 3) | (((hw >> 8) & 15) << 2) | (((hw >> 12) & 1) << 6) | (((hw >> 13) & 7) << 7) | (((hw >> 32) & 15) << 10)
    u, cnt = np.unique(simd, return_counts=True)
    busy = np.array([life[simd == s].sum() for s in u]) / span
    hs = np.zeros(10)
    for sid in u[::16]:
        m = simd == sid
        e = np.concatenate([np.stack([t0[m], np.ones(m.sum(), dtype=np.int64)], 1), np.stack([t1[m], -np.ones(m.sum(), dtype=np.int64)], 1)]); e = e[np.argsort(e[:, 0], kind="stable")]
        l_ = np.cumsum(e[:, 1]); d_ = np.diff(e[:, 0], append=e[-1, 0])
        for c in range(10): hs[c] += d_[l_ == c].sum()
        hs[0] += (e[0, 0] - base) + (t1.max() - e[-1, 0])
    print("   per SIMD (64 of them): share of time with 0, 1, 2, .. resident trace waves:", np.round(hs / hs.sum(), 3).tolist())
    print(f"   distinct SIMD ids {len(u)}; waves per SIMD id min {cnt.min()} mean {cnt.mean():.1f} max {cnt.max()}; resident waves per SIMD id: min {busy.min():.2f} mean {busy.mean():.2f} max {busy.max():.2f}")
rp.close()
