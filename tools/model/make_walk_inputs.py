"""Inputs of the host walk models (grid_start_model.cpp, persistent_lanes_model.cpp, queries_per_lane_model.cpp): bench pair seed 10 as float
triples (tgt.f32, src.f32) and the transforms of the first sweeps (poses.f32: identity, then the CPU path's iterates), in /tmp/wm (source in
input order) and /tmp/wms (source sorted along its own Morton curve).  Builder-side analysis tool (docs/NOTEBOOK_r5.md section 1)."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from locus_amd import synth
from oracle import oracle as O
src, tgt, delta = synth.scan_pair(n_rings=64, n_az=1563, scale=2.0, noise=0.02, seed=10)
s4, t4 = O.xyz4(src), O.xyz4(tgt)
t0=time.time()
ns = O.normals_knn(s4, 20, threads=32); nt = O.normals_knn(t4, 20, threads=32)
print('normals', time.time()-t0)
P = O.default_params(max_iterations=20, max_inner_iterations=20, corr_dist=1.0, transformation_epsilon=1e-12, rotation_epsilon=1e-12, num_threads=32)
r = O.gicp_align(s4, O.nrm4(ns[:, :3]) if ns.shape[1]>=3 else ns, t4, O.nrm4(nt[:, :3]), P)
print(r['iterations'], time.time()-t0)
Ts = r['trace']['T']
poses = [np.eye(4, dtype=np.float32)[:3].copy()]
for k in range(4):
    M = Ts[k].reshape(4,4).T
    poses.append(M[:3].astype(np.float32))
poses = np.stack(poses).astype(np.float32)
def morton_order(p):
    lo, hi = p.min(0), p.max(0)
    sc = 1023.999/ (hi-lo).max()
    q = np.clip(((p-lo)*sc).astype(np.int64), 0, 1023)
    def ex(v):
        v = v & 0x3ff
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    key = (ex(q[:,0])<<2)|(ex(q[:,1])<<1)|ex(q[:,2])
    return np.argsort(key, kind='stable')
import os
os.makedirs('/tmp/wm', exist_ok=True); os.makedirs('/tmp/wms', exist_ok=True)
for d, order in (('/tmp/wm', np.arange(len(src))), ('/tmp/wms', morton_order(src))):
    src[order].astype(np.float32).tofile(d+'/src.f32'); tgt.astype(np.float32).tofile(d+'/tgt.f32'); poses.tofile(d+'/poses.f32')
print('done')
