import sys, os, gzip, json, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, 'tests')
import sedef_amd
from oracle.binding import Oracle, cigar_to_str
from util import codes
cases = json.loads(gzip.open('tests/golden/extz2_kat.json.gz').read())['cases']
c = [c for c in cases if c['tag']=='grid' and len(c['q'])==16 and c['w']==15][0]
q, t = codes(c['q']), codes(c['t']); w = c['w']
qlen, tlen = len(q), len(t)
eng = sedef_amd.Extz2Engine(0)
lib = eng.lib
lib.sdf_debug_copy_dir.argtypes=[C.c_void_p, C.c_void_p, C.c_size_t]
def band(r):
    lo=max(0,r-qlen+1,(r-w+1)>>1); hi=min(tlen-1,r,(r+w)>>1)
    return lo,hi,lo//16*16,(hi+16)//16*16-1
nrow=qlen+tlen-1
ncol16=((min(qlen,tlen,w+1)+15)//16+1)*16
# general
res,cig = eng.align_pairs([(q,t)], w=w, want=7)
buf=np.zeros(nrow*ncol16+64,np.uint8); lib.sdf_debug_copy_dir(eng.ctx, buf.ctypes.data, buf.nbytes)
G={}
for r in range(nrow):
    lo0,hi0,lo,hi=band(r)
    for tt in range(lo,hi+1): G[(r,tt)]=int(buf[r*ncol16+tt-lo])
print('general cigar', cigar_to_str(cig[:res[0]['n_cigar']]), c['expect']['cigar'])
res,cig = eng.align_pairs([(q,t)], w=w, want=3)
print('wave cigar', cigar_to_str(cig[:res[0]['n_cigar']]))
nblk=(nrow+15)//16
wb=np.zeros(nblk*64*4,np.uint32); lib.sdf_debug_copy_dir(eng.ctx, wb.ctypes.data, wb.nbytes)
wb=wb.reshape(nblk,64,4)
bad=0
for r in range(nrow):
    lo0,hi0,lo,hi=band(r)
    base=band(r//16*16)[2]
    for tt in range(lo,hi+1):
        slot=tt-base; lane=slot>>1; bit=15-(r&15)+16*(slot&1)
        fa,fb,fx,fy=[(int(wb[r//16,lane,f])>>bit)&1 for f in range(4)]
        d=(2 if fb else fa)|fx<<3|fy<<4
        if d!=G[(r,tt)]:
            bad+=1
            if bad<40: print('diff r=%d t=%d (lo0=%d hi0=%d lo=%d hi=%d) general=%02x wave=%02x'%(r,tt,lo0,hi0,lo,hi,G[(r,tt)],d))
print('bad',bad)
