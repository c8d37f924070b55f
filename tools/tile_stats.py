import sys; sys.path.insert(0,'.')
import numpy as np
from godotgaussiansplatting_amd import capi, scenes
import bench
n,deg,w,h,seed,vp,cam=bench.build_scene_inputs('c3')
ctx=capi.Context(n,w,h)
from godotgaussiansplatting_amd import scenes
_rows=scenes.config_rows('c3')
for _f in range(0,n,1<<20): ctx.upload_ply_rows(_rows[_f:_f+(1<<20)],first=_f,load_time=-10.0)
fr=capi.make_frame(vp,cam); ctx.render(fr); ctx.synchronize()
st=ctx.read_tile_staged().astype(np.int64); b=ctx.read_bounds().astype(np.int64); nt=np.clip(b[:,1]-b[:,0],0,None)
print('staged sum',st.sum(),'max',st.max(),'p99',np.percentile(st,99),'p90',np.percentile(st,90),'mean',st.mean())
batches=(st+255)//256
print('batches max',batches.max(),'hist',np.bincount(batches)[:45])
gx=120
hm=batches.reshape(68,120)
print('rows batches sum', hm.sum(1))
