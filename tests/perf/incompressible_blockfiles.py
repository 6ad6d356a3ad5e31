#!/usr/bin/env python3
"""Block files of flags that hardly compress (12-bit and 16-bit uniform) through both decoders of both codecs: where the decoded
bytes are no more than the compressed ones the host-thread pipeline is PCIe-bound at its best and the GPU decoders can only lose."""
import sys, os, time, ctypes, struct
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tools'); sys.path.insert(0, ROOT+'/tests/perf')
import numpy as np, blockfile_tool as bt, oracle
from libflagstats_amd import _lib
from concurrent.futures import ThreadPoolExecutor
lib=_lib.lib(); _lib.check(lib.FLAGSTATS_hip_init(0),"init")
n=2**28
for codec, mode, level in (("zstd", "zstd", 1), ("lz4", "fast", 2)):
  knob = b"zstd_decoder" if codec == "zstd" else b"lz4_decoder"
  entry = lib.FLAGSTATS_hip_blockimage_zstd if codec == "zstd" else lib.FLAGSTATS_hip_blockimage_lz4
  for name,mask in (("12-bit uniform",0x0FFF),("16-bit uniform",0xFFFF)):
    per=512000
    def make(i):
        f=oracle.generate(oracle.GEN_UNIFORM,3,mask,i*per,min(per,n-i*per))
        c=bt.compress_block(f.tobytes(),mode,level)
        return struct.pack("<ii",f.nbytes,len(c))+c
    with ThreadPoolExecutor(16) as ex: img=b"".join(ex.map(make,range((n+per-1)//per)))
    buf=np.frombuffer(img,dtype=np.uint8)
    res={}
    for dec in (0,1,2):
        _lib.check(lib.FLAGSTATS_hip_set(knob,dec),"set")
        ts=[]
        for rep in range(3):
            out=np.zeros(32,dtype=np.uint64); st=_lib.BlockfileStats()
            t0=time.perf_counter()
            _lib.check(entry(buf.ctypes.data,buf.size,0,out.ctypes.data,ctypes.byref(st)),"img")
            ts.append(time.perf_counter()-t0)
        res[dec]=(min(ts), st.gpu_decode)
    print("%s, %s: %d MiB compressed (ratio %.2f): host threads %.1f ms, GPU decode %.1f ms, by the default rule %.1f ms (%s)" % (codec,name,len(img)>>20,2*n/len(img),res[0][0]*1e3,res[1][0]*1e3,res[2][0]*1e3,"GPU" if res[2][1] else "host"),flush=True)
  _lib.check(lib.FLAGSTATS_hip_set(knob,2),"set")
