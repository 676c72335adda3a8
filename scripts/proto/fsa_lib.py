theta=0.02; omega=0.8; omega_p=0.66
def Tm(d):
    T=np.tile(np.eye(3),(d.shape[0],1,1)); T[:,0,2]=-d[:,1]; T[:,1,2]=d[:,0]; return T
def aggregate(n, indptr, indices, w, wd, theta):
    agg=-np.ones(n,dtype=int); S=[]
    for i in range(n):
        ks=np.arange(indptr[i],indptr[i+1]); js=indices[ks]
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
        js,ws=S[i]; m=agg1[js]>=0
        if m.any(): agg[i]=agg1[js[m][np.argmax(ws[m])]]
    for i in range(n):
        if agg[i]>=0: continue
        agg[i]=nc; js,_=S[i]
        for j in js:
            if agg[j]<0: agg[j]=nc
        nc+=1
    return agg,nc
def build(H,pos0,reuse=None,dynamic_mask=True):
    levels=[]; A=H.tobsr(blocksize=(3,3)); pos=pos0; l=0
    while True:
        n=A.shape[0]//3; A.sort_indices()
        indptr,indices,data=A.indptr,A.indices,A.data
        rows=np.repeat(np.arange(n),np.diff(indptr)); dm=rows==indices
        D=np.zeros((n,3,3)); D[rows[dm]]=data[dm]
        Dinv_m=sp.bsr_matrix((np.linalg.inv(0.5*(D+D.transpose(0,2,1))),np.arange(n),np.arange(n+1)),shape=(3*n,3*n)).tocsr()
        lev=dict(A=A.tocsr(),Dinv=Dinv_m,n=n); levels.append(lev)
        last = (n<=400) if reuse is None else ('agg' not in reuse[l])
        if last:
            lev['lu']=spla.splu(A.tocsc()); break
        w=np.sqrt((data**2).sum(axis=(1,2))); wd=np.zeros(n); wd[rows[dm]]=w[dm]
        if reuse is None:
            agg,nc=aggregate(n,indptr,indices,w,wd,theta)
            if nc>0.9*n: agg,nc=aggregate(n,indptr,indices,w,wd,0.0)
        else:
            agg,nc=reuse[l]['agg'],reuse[l]['nc']
        lev['agg']=agg; lev['nc']=nc
        cent=np.zeros((nc,2)); np.add.at(cent,agg,pos); cent/=np.bincount(agg,minlength=nc)[:,None]
        T=sp.bsr_matrix((Tm(pos-cent[agg]),agg,np.arange(n+1)),shape=(3*n,3*nc)).tocsr()
        strong=(w*w>=theta*theta*wd[rows]*wd[indices])|dm
        if reuse is not None and dynamic_mask is not True:
            # the stale mask by (row, col) key (patterns of coarse levels may differ slightly: default weak)
            key=rows.astype(np.int64)*n+indices
            old=reuse[l]['strong_keys']
            fresh_strong=strong
            if dynamic_mask=="and": strong=(np.isin(key,old)&strong)|dm      # pattern-restricted dynamic mask
            else: strong=np.isin(key,old)|dm
            nolump = fresh_strong & ~strong if dynamic_mask=="nolump" else None   # newly strong, not in the pattern: kept out of the lumping
        lev['strong_keys']=(rows.astype(np.int64)*n+indices)[strong]
        weak=~strong
        if reuse is not None and dynamic_mask=="nolump": weak=weak & ~nolump
        G=Tm(pos[indices[weak]]-pos[rows[weak]])
        corr=np.zeros((n,3,3)); np.add.at(corr,rows[weak],data[weak]@G)
        DF=D+corr
        AF=sp.bsr_matrix((data[strong].copy(),indices[strong],np.concatenate([[0],np.cumsum(np.bincount(rows[strong],minlength=n))])),shape=A.shape)
        dmk=(np.repeat(np.arange(n),np.diff(AF.indptr))==AF.indices); AF.data[dmk]=DF
        DFinv_m=sp.bsr_matrix((np.linalg.inv(DF),np.arange(n),np.arange(n+1)),shape=(3*n,3*n)).tocsr()
        P=(T-omega_p*DFinv_m@(AF.tocsr()@T)).tocsr()
        lev['P']=P
        A=(P.T@lev['A']@P).tobsr(blocksize=(3,3)); pos=cent; l+=1
    return levels
def cyc(levels,l,r):
    L=levels[l]
    if 'lu' in L: return L['lu'].solve(r)
    x=omega*(L['Dinv']@r); rr=r-L['A']@x
    x=x+L['P']@cyc(levels,l+1,L['P'].T@rr); rr=r-L['A']@x
    return x+omega*(L['Dinv']@rr)
def solve(levels,b,maxit=1500):
    Hc=levels[0]['A']; x=np.zeros_like(b); r=b.copy(); z=cyc(levels,0,r); p=z.copy(); rz=r@z; bn=np.linalg.norm(b); it=0
    while it<maxit:
        q=Hc@p; a=rz/(p@q); x+=a*p; r-=a*q; it+=1
        if np.linalg.norm(r)<=1e-8*bn: break
        z=cyc(levels,0,r); rzn=r@z; p=z+(rzn/rz)*p; rz=rzn
    return it,x
