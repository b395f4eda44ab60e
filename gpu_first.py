# scratch: first contact with the GPU (smoke + parity sample + timing); superseded by tests/ and bench.py
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import __graft_entry__ as g
import zstd_amd
from zstd_amd import corpus
t0=time.time(); g.smoke(); print('smoke %.2fs'%(time.time()-t0), flush=True)
N=int(sys.argv[1]) if len(sys.argv)>1 else 1024
FS=131072
host=corpus.generate('log',FS,N,42,threads=16)
src=torch.from_numpy(host).cuda()
bound=zstd_amd.CompressBound(FS)
stride=(bound+255)//256*256
dst=torch.empty(N*stride,dtype=torch.uint8,device='cuda')
b=zstd_amd.DeviceBatch(N)
st=torch.cuda.current_stream().cuda_stream
b.set_frames(src.data_ptr(),[i*FS for i in range(N)],[FS]*N,dst.data_ptr(),[i*stride for i in range(N)],[bound]*N,st)
for it in range(2):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); b.compress(1,None,st); e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1); print('compress %d frames: %.2f ms  %.2f GB/s  launch %s'%(N,ms,N*FS/ms/1e6,b.last_launch()),flush=True)
cs=b.results(st)
bad=[c for c in cs if c>2**63]; print('errors',len(bad),'ratio %.3f'%(N*FS/sum(cs)))
# verify vs oracle on a sample
orc=ctypes.CDLL('oracle/liboracle.so'); orc.zo_compress.restype=ctypes.c_size_t
orc.zo_compress.argtypes=[ctypes.c_void_p,ctypes.c_size_t,ctypes.c_char_p,ctypes.c_size_t,ctypes.c_int]
dh=dst.cpu().numpy()
nbad=0
for i in range(0,N,max(1,N//64)):
    buf=ctypes.create_string_buffer(bound+1); n=orc.zo_compress(buf,bound,host[i*FS:(i+1)*FS].tobytes(),FS,1)
    if buf.raw[:n]!=dh[i*stride:i*stride+cs[i]].tobytes(): nbad+=1
print('oracle mismatches on sample:',nbad,flush=True)
# decompress
out=torch.empty(N*FS,dtype=torch.uint8,device='cuda')
b2=zstd_amd.DeviceBatch(N)
b2.set_frames(dst.data_ptr(),[i*stride for i in range(N)],cs,out.data_ptr(),[i*FS for i in range(N)],[FS]*N,st)
for it in range(2):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); b2.decompress(None,st); e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1); print('decompress %d frames: %.2f ms  %.2f GB/s launch %s'%(N,ms,N*FS/ms/1e6,b2.last_launch()),flush=True)
ds=b2.results(st)
print('decode errors',len([d for d in ds if d!=FS]),'roundtrip equal',bool(torch.equal(out,src)))
