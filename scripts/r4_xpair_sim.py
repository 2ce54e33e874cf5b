# Lane-pair SHA-256 (the express form, k_sha256_xpair in pbs_plus_amd/csrc/kernels.hip), checked lane by lane against hashlib before
# any kernel was written: lane A carries e,f,g,h, lane B a,b,c,d two slot-rounds behind; every slot is ONE operation on both lanes.
import hashlib, struct, os
M=0xffffffff
K=[0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2]
IV=[0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19]
rotr=lambda x,n:((x>>n)|(x<<(32-n)))&M
def sched(block):
    W=list(struct.unpack('>16I',block))
    for t in range(16,64):
        s0=rotr(W[t-15],7)^rotr(W[t-15],18)^(W[t-15]>>3); s1=rotr(W[t-2],17)^rotr(W[t-2],19)^(W[t-2]>>10)
        W.append((W[t-16]+s0+W[t-7]+s1)&M)
    return [(W[t]+K[t])&M for t in range(64)]
SH={'A':(6,11,25),'B':(2,13,22)}; ROLE={'A':0,'B':M}
def block_pair(HA,HB,KW):
    """HA = [H4..H7] (lane A), HB = [H0..H3] (lane B); returns new (HA, HB). Lock-step: every slot is the same op on both lanes."""
    sel=lambda lane,a,b: b if lane=='B' else a         # per-lane select (v_cndmask / bfi with ROLE)
    X={l:[None]*4 for l in 'AB'}
    HR={'A':HA,'B':HB}
    for l in 'AB':
        X[l][0]=sel(l,HR[l][0],HR[l][2]); X[l][1]=sel(l,HR[l][1],HR[l][3]); X[l][2]=HR[l][2]; X[l][3]=HR[l][3]
    out={l:[] for l in 'AB'}
    for r in range(66):
        kw=KW[min(r,63)]
        new={}
        # snapshot for the DPP read (all lanes read the OLD register contents)
        X1={l:X[l][1] for l in 'AB'}
        for l,o in (('A','B'),('B','A')):
            x0,x1,x2,x3=X[l]
            r1=rotr(x0,SH[l][0]); r2=rotr(x0,SH[l][1]); r3=rotr(x0,SH[l][2])
            S=r1^r2^r3
            SEL=x0^(x2&ROLE[l])
            F=(SEL&x1)|(~SEL&x2&M)
            CKW=kw if l=='A' else 1
            NZ=((x3^ROLE[l])+CKW)&M
            P=(X1[o]+NZ)&M
            new[l]=(S+F+P)&M
        for l in 'AB':
            v=new[l]
            if l=='B' and r<2: v=HR['B'][1-r]          # select: B's chain starts two slot-rounds late
            out[l].append(v)
            X[l]=[v,X[l][0],X[l][1],X[l][2]]
    fin={'A':[out['A'][63],out['A'][62],out['A'][61],out['A'][60]],'B':[out['B'][65],out['B'][64],out['B'][63],out['B'][62]]}
    return [ (HA[i]+fin['A'][i])&M for i in range(4)], [(HB[i]+fin['B'][i])&M for i in range(4)]
def sha(msg):
    ml=len(msg); m=msg+b'\x80'+b'\0'*((55-ml)%64)+struct.pack('>Q',ml*8)
    HA=IV[4:]; HB=IV[:4]
    for i in range(0,len(m),64): HA,HB=block_pair(HA,HB,sched(m[i:i+64]))
    return struct.pack('>8I',*(HB+HA))
for n in [0,1,55,56,63,64,65,1000,4096]:
    d=os.urandom(n); assert sha(d)==hashlib.sha256(d).digest(), n
print("lane-pair SHA-256 formulation matches hashlib")
