import ctypes, os, struct, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tools')
import numpy as np, blockfile_tool as bt, oracle
from libflagstats_amd import _lib
lib=_lib.lib(); _lib.check(lib.FLAGSTATS_hip_init(0),"init")
n=2**28; per=512000
def gen(kind,i):
    if kind=="constant": return np.full(per, 99, dtype=np.uint16)
    if kind=="period7": return np.tile(np.array([99,147,83,163,1187,1107,0],dtype=np.uint16), per//7+1)[:per].copy()
    period=int(kind[6:])   # "period2000": 2000 random flags repeated (periods of 20,000 / 40,000 flags = 40 / 80 KB lie beyond the
    r=np.random.default_rng(i); a=r.integers(0,4096,period,dtype=np.uint16); return np.tile(a, per//period+1)[:per].copy()   # 32 KiB the Zstandard kernel's ring keeps)
for codec,mode,level in (("zstd","zstd",1),("lz4","fast",2)):
    knob=b"zstd_decoder" if codec=="zstd" else b"lz4_decoder"
    entry=lib.FLAGSTATS_hip_blockimage_zstd if codec=="zstd" else lib.FLAGSTATS_hip_blockimage_lz4
    for kind in ("constant","period7","period2000","period10000","period20000","period40000"):
        def make(i):
            f=gen(kind,i); c=bt.compress_block(f.tobytes(),mode,level); return struct.pack("<ii",f.nbytes,len(c))+c
        with ThreadPoolExecutor(16) as ex: img=b"".join(ex.map(make,range(n//per)))
        buf=np.frombuffer(img,dtype=np.uint8); res={}
        want=None
        for dec in (0,1):
            _lib.check(lib.FLAGSTATS_hip_set(knob,dec),"set"); ts=[]
            for rep in range(3):
                out=np.zeros(32,dtype=np.uint64); st=_lib.BlockfileStats(); t0=time.perf_counter()
                _lib.check(entry(buf.ctypes.data,buf.size,0,out.ctypes.data,ctypes.byref(st)),"image"); ts.append(time.perf_counter()-t0)
            if want is None: want=out.copy()
            assert np.array_equal(out,want)
            res[dec]=min(ts)
        print("%s, %s: ratio %.0f (%.1f MiB): host threads %.1f ms, GPU decode %.1f ms"%(codec,kind,2.0*(n//per)*per/len(img),len(img)/2**20,res[0]*1e3,res[1]*1e3),flush=True)
    _lib.check(lib.FLAGSTATS_hip_set(knob,2),"set")
