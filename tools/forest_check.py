import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch

from stratum_amd import camera, scenes, wire
from stratum_amd.bdpt import BDPT
from oracle import oracle_py as orc
t=time.time(); sc, cam = scenes.forest(); print("gen %.1fs tris %d inst %d verts %d" % (time.time()-t, sc.triangle_count, sc.instances.shape[0], sc.vertices.shape[0]), flush=True)
r = BDPT(0, args={"maxDiffuseVertices": 8, "maxPathVertices": 10, "minPathVertices": 4, "bdptFlag": ["~coherentrr"]})
t=time.time(); r.update(sc); print("upload+build %.1fs" % (time.time()-t), flush=True)
fr = camera.Frame(320, 180, cam["fovy"], cam["eye"], cam["target"])
got = r.render(fr, 0, 1)
t=time.time(); o = orc.OracleScene(sc); print("oracle build %.1fs" % (time.time()-t), flush=True)
t=time.time(); ref = o.render(fr, r.push_constants(fr), r.mSamplingFlags, 0, 1, threads=16); print("oracle render %.1fs" % (time.time()-t), flush=True)
print("vis equal", np.array_equal(got["visibility"]["instance_primitive_index"], ref["visibility"]["instance_primitive_index"]), "rays", got["ray_count"], ref["ray_count"])
a=got["radiance"][...,:3].astype(np.float64); b=ref["radiance"][...,:3].astype(np.float64)
print("rel-L2 %.3e" % (np.sqrt(((a-b)**2).sum())/np.sqrt((b**2).sum())), "bit-diff pixels", int((got["radiance"].view(np.uint32)!=ref["radiance"].view(np.uint32)).any(-1).sum()))
# timing at 4K

fr4 = camera.Frame(3840, 2160, cam["fovy"], cam["eye"], cam["target"])
rad = torch.zeros((2160,3840,4), device="cuda"); out={"radiance": rad.data_ptr()}
import os
if os.environ.get("STHIP_FUSE_TRACE"): r.set_option("fuse_trace", int(os.environ["STHIP_FUSE_TRACE"]))
for i in range(2): r.render(fr4, i, 1, device_outputs=out)
torch.cuda.synchronize(); t=time.time()
for i in range(4): r.render(fr4, 2+i, 1, device_outputs=out)
torch.cuda.synchronize(); dt=(time.time()-t)/4
st = r.stats(); print("4K: %.2f ms/frame, rays %d -> %.1f Mray/s" % (dt*1e3, st["rays_total"], st["rays_total"]/dt/1e6))
from PIL import Image
img = got["radiance"][...,:3]; ldr=np.clip(img/(1+img),0,1)**(1/2.2)
Image.fromarray((ldr*255).astype(np.uint8)).save("/root/repo/gpurun_out/forest.png")
