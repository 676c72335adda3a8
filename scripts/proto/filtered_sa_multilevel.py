import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import np_oracle
from sparse_gslam_amd import synth

V=int(sys.argv[1]); E=int(sys.argv[2]); mode=sys.argv[3] if len(sys.argv)>3 else "filtered"
theta=0.02; omega=0.8; omega_p=0.66
g=synth.manhattan(V,E,seed=4,init="odom")
H,b,c2,rc2=np_oracle.linearize(g.poses,g.fixed,g.ei,g.ej,g.meas,g.info,g.phi)
free=np.flatnonzero(~g.fixed); pos0=g.poses[free,:2]

def Tm(d):
    T=np.tile(np.eye(3),(d.shape[0],1,1)); T[:,0,2]=-d[:,1]; T[:,1,2]=d[:,0]; return T

def aggregate(n, indptr, indices, w, wd, theta):
    strong_lists=[]
    agg=-np.ones(n,dtype=int)
    S=[]
    for i in range(n):
        ks=np.arange(indptr[i],indptr[i+1])
        js=indices[ks]
        m=(js!=i)&(w[ks]**2>=theta*theta*wd[i]*wd[js])&(w[ks]>0)
        S.append((js[m],w[ks][m]))
    nc=0
    for i in range(n):
        js,_=S[i]
        if agg[i]>=0 or len(js)==0: continue
        if (agg[js]>=0).any(): continue
        agg[i]=nc; agg[js]=nc; nc+=1
    agg1=agg.copy()
    for i in range(n):
        if agg1[i]>=0: continue
        js,ws=S[i]
        m=agg1[js]>=0
        if m.any():
            agg[i]=agg1[js[m][np.argmax(ws[m])]]
    for i in range(n):
        if agg[i]>=0: continue
        agg[i]=nc
        js,_=S[i]
        for j in js:
            if agg[j]<0: agg[j]=nc
        nc+=1
    return agg,nc

levels=[]
A=H.tobsr(blocksize=(3,3)); pos=pos0
while True:
    n=A.shape[0]//3
    A.sort_indices()
    indptr,indices,data=A.indptr,A.indices,A.data
    rows=np.repeat(np.arange(n),np.diff(indptr))
    dm=rows==indices
    D=np.zeros((n,3,3)); D[rows[dm]]=data[dm]; D=0.5*(D+D.transpose(0,2,1))
    Dinv=np.linalg.inv(D)
    Dinv_m=sp.bsr_matrix((Dinv,np.arange(n),np.arange(n+1)),shape=(3*n,3*n)).tocsr()
    lev=dict(A=A.tocsr(),Dinv=Dinv_m,n=n)
    levels.append(lev)
    if n<=400: 
        lev['lu']=spla.splu(A.tocsc()); break
    w=np.sqrt((data**2).sum(axis=(1,2)))
    wd=np.zeros(n); wd[rows[dm]]=w[dm]
    agg,nc=aggregate(n,indptr,indices,w,wd,theta)
    if nc>0.9*n:
        agg,nc=aggregate(n,indptr,indices,w,wd,0.0)
    cent=np.zeros((nc,2)); np.add.at(cent,agg,pos); cent/=np.bincount(agg,minlength=nc)[:,None]
    T=sp.bsr_matrix((Tm(pos-cent[agg]),agg,np.arange(n+1)),shape=(3*n,3*nc)).tocsr()
    strong=(w*w>=theta*theta*wd[rows]*wd[indices])|dm
    # rigid-motion row sums: sum_j A_ij T(p_j - p_i) should vanish in the interior
    Gall=Tm(pos[indices]-pos[rows]); rs_=np.zeros((n,3,3)); np.add.at(rs_,rows,data@Gall)
    rel=np.sqrt((rs_**2).sum(axis=(1,2)))/np.sqrt((D**2).sum(axis=(1,2)))
    print(f"   level {len(levels)-1}: rigid row-sum residual rel to |D|: median {np.median(rel):.2e} 99% {np.quantile(rel,0.99):.2e} max {rel.max():.2e}")
    kind="tentative"
    P=T
    if mode!="tentative":
        # unfiltered first
        Pu=(T-omega_p*Dinv_m@(lev['A']@T)).tocsr()
        Acu=(Pu.T@lev['A']@Pu)
        if mode=="unfiltered" or Acu.nnz//9 <= 1.5*max(A.nnz//9,4096)*0+ (A.nnz//9)*1.0 and Pu.nnz//9 <= 8*n and mode=="auto":
            P=Pu; kind="smoothed"
        else:
            keep=strong; weak=~strong
            G=Tm(pos[indices[weak]]-pos[rows[weak]])
            corr=np.zeros((n,3,3)); np.add.at(corr,rows[weak],data[weak]@G)
            DF=D+corr
            AF=sp.bsr_matrix((data[keep].copy(),indices[keep],np.concatenate([[0],np.cumsum(np.bincount(rows[keep],minlength=n))])),shape=A.shape)
            dmk=(np.repeat(np.arange(n),np.diff(AF.indptr))==AF.indices)
            AF.data[dmk]=DF
            DFinv_m=sp.bsr_matrix((np.linalg.inv(DF),np.arange(n),np.arange(n+1)),shape=(3*n,3*n)).tocsr()
            P=(T-omega_p*DFinv_m@(AF.tocsr()@T)).tocsr(); kind="filtered"
    Ac=(P.T@lev['A']@P).tobsr(blocksize=(3,3))
    lev['P']=P
    print(f"level {len(levels)-1}: n={n} nnzb={A.nnz//9} -> nc={nc} ({kind}, P {P.nnz/9/n:.2f}/row, weak frac {(~strong).sum()/max(1,(~dm).sum()):.2f}) coarse nnzb={Ac.nnz//9}",flush=True)
    A=Ac; pos=cent

KD=int(sys.argv[4]) if len(sys.argv)>4 else 0      # levels 1..KD solved by FCG steps (K-cycle); 0 = V-cycle
F2=int(sys.argv[5]) if len(sys.argv)>5 else 1      # levels <= F2 take two FCG steps, deeper K levels one
def cyc(l,r):
    L=levels[l]
    if 'lu' in L: return L['lu'].solve(r)
    x=omega*(L['Dinv']@r)
    rr=r-L['A']@x
    x=x+L['P']@coarse(l+1,L['P'].T@rr)
    rr=r-L['A']@x
    return x+omega*(L['Dinv']@rr)
def coarse(l,r):
    L=levels[l]
    if 'lu' in L: return L['lu'].solve(r)
    if l>KD: return cyc(l,r)
    A=L['A']
    z1=cyc(l,r); q1=A@z1; a1=(z1@r)/(z1@q1)
    if l>F2: return a1*z1
    r2=r-a1*q1
    z2=cyc(l,r2); 
    beta=(z2@q1)/(z1@q1)
    p2=z2-beta*z1; q2=A@p2
    a2=(p2@r2)/(p2@q2)
    return a1*z1+a2*p2
def vcycle(l,r): return cyc(l,r)
Hc=levels[0]['A']
x=np.zeros_like(b); r=b.copy(); z=vcycle(0,r); p=z.copy(); rz=r@z; bn=np.linalg.norm(b); it=0
hist=[]
while it<3000:
    q=Hc@p; a=rz/(p@q); x+=a*p; r-=a*q; it+=1
    rn=np.linalg.norm(r)/bn; hist.append(rn)
    if rn<=1e-8: break
    r_old=r+a*q; z=vcycle(0,r); rzn=r@z; p=z+((z@(r-r_old))/rz)*p; rz=rzn
print(f"mode {mode} KD={KD} F2={F2}: PCG iterations {it}; relres after 10/20/50/100: "+" ".join(f"{hist[min(k,len(hist)-1)]:.1e}" for k in (9,19,49,99)))

# diagnostics: quality of the cycle on every level as a preconditioner for THAT level's operator (random right-hand side),
# with the exact solve below (two-level) and with the recursive cycle below (multilevel)
rng=np.random.default_rng(0)
def pcg_level(l, prec, maxit=2000):
    A=levels[l]['A']; bb=rng.standard_normal(A.shape[0]); x=np.zeros_like(bb); r=bb.copy(); z=prec(r); p=z.copy(); rz=r@z; it=0; bn=np.linalg.norm(bb)
    while it<maxit:
        q=A@p; a=rz/(p@q); x+=a*p; r-=a*q; it+=1
        if np.linalg.norm(r)<=1e-8*bn: break
        z=prec(r); rzn=r@z; p=z+(rzn/rz)*p; rz=rzn
    return it
KD=0
for l in range(len(levels)-1):
    L=levels[l]
    lu=spla.splu((levels[l+1]['A']).tocsc())
    def two(r,L=L,lu=lu):
        x=omega*(L['Dinv']@r); rr=r-L['A']@x; x=x+L['P']@lu.solve(L['P'].T@rr); rr=r-L['A']@x; return x+omega*(L['Dinv']@rr)
    print(f"level {l}: two-level {pcg_level(l,two)} its; multilevel V {pcg_level(l,lambda r,l=l: cyc(l,r))} its",flush=True)
