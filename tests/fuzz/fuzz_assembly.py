"""Random mapped patches through the FE-side assembly (csrc/tg_assemble.hip: mass, Laplace, nodal load; the dolfin.assemble
stand-in of SURVEY 8f-1, tIGAr/common.py:917-945, 1206-1220) against ``oracle.mapped_fe_system`` (developer tool): dimension
1-3 embedded in 1-3 space dimensions, degrees 1-4, non-uniform element sizes, perturbed and rational geometries, Gauss
points p+1 / p+2.  Where nsd == d also two field blocks of the elasticity form (one and its transposed partner) against
``oracle.mapped_elasticity_fe_system`` and the biharmonic form against ``oracle.mapped_biharmonic_fe_system`` (round 6).

    python tests/fuzz/fuzz_assembly.py [cases]"""
import sys, json, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import tigar_oracle as O
from tigar_amd import device as dev
bad=0
N=int(sys.argv[1]) if len(sys.argv)>1 else 60
for i in range(N):
    rng=np.random.default_rng([77,i])
    d=int(rng.choice([1,2,2,3])); p=int(rng.integers(1,5 if d<3 else 4))
    nsd=int(rng.integers(d, 4))
    nmax={1:12,2:6,3:3}[d]
    uks=[]
    for k in range(d):
        n=int(rng.integers(1,nmax+1))
        br=np.concatenate([[0.0],np.sort(rng.uniform(0.1,0.9,n-1)),[1.0]])*(1+k)
        if n>1 and np.min(np.diff(br))<0.02: br=np.linspace(0,1+k,n+1)
        uks.append(br)
    # FE nodes (degree p, equally spaced inside each element), direction 0 fastest
    axes=[]
    for u in uks:
        a=[u[0]]
        for e in range(len(u)-1):
            a+= [u[e]+(u[e+1]-u[e])*j/p for j in range(1,p+1)]
        axes.append(np.array(a))
    grids=np.meshgrid(*axes, indexing='ij')
    X=[g.transpose(*reversed(range(d))).ravel() for g in grids]     # direction 0 fastest
    w=1.0+0.2*np.sin(sum((k+1.3)*x for k,x in enumerate(X)))*(rng.random()<0.5)
    cp=[]
    for c in range(nsd):
        base=X[c] if c<d else 0.3*np.sin(1.7*X[0])*(np.cos(0.9*X[d-1]) if d>1 else 1.0)
        pert=0.08*np.sin(2.1*X[(c+1)%d]+0.4*c)*np.cos(1.3*X[0])
        cp.append((base+pert)*w)
    cp.append(np.asarray(w)*np.ones_like(X[0]))
    nq=int(rng.choice([0,p+1,p+2]))
    case=dict(i=i,d=d,p=p,nsd=nsd,n=[len(u)-1 for u in uks],nq=nq)
    try:
        fn=np.cos(X[0])*(1+X[d-1])
        Mo,Ko,bo=O.mapped_fe_system(uks,p,cp,nq=(nq or None),fnodal=fn)
        cpd=[dev.DeviceVector(data=np.ascontiguousarray(c)) for c in cp]
        for form,Ao in (("mass",Mo),("laplace",Ko)):
            A=dev.assemble_mapped_matrix(uks,p,cpd,form,nq=(nq or None)).to_scipy()
            assert A.shape==Ao.shape
            e=abs(A-Ao).max()/abs(Ao).max()
            assert e<=1e-11, "%s: %g"%(form,e)
        b=dev.assemble_mapped_load(uks,p,cpd,dev.DeviceVector(data=fn),nq=(nq or None)).get_local()
        e=np.max(np.abs(b-bo))/np.max(np.abs(bo))
        assert e<=1e-12, "load: %g"%e
        if nsd==d:
            lam,mu=float(rng.uniform(0.2,3.0)),float(rng.uniform(0.2,2.0))
            Eo=O.mapped_elasticity_fe_system(uks,p,cp,lam,mu,nq=(nq or None))
            Nn=Eo.shape[0]//d; sc=abs(Eo).max()
            bi,bj=int(rng.integers(0,d)),int(rng.integers(0,d))
            Bij=dev.assemble_mapped_elasticity_block(uks,p,cpd,bi,bj,lam,mu,nq=(nq or None)).to_scipy()
            Bji=dev.assemble_mapped_elasticity_block(uks,p,cpd,bj,bi,lam,mu,nq=(nq or None)).to_scipy()
            e=abs(Bij-Eo[bi*Nn:(bi+1)*Nn,bj*Nn:(bj+1)*Nn]).max()/sc
            assert e<=1e-11, "elasticity block (%d,%d): %g"%(bi,bj,e)
            e=abs(Bij-Bji.T).max()/sc
            assert e<=1e-11, "elasticity block (%d,%d) against the transposed partner: %g"%(bi,bj,e)
            Ho=O.mapped_biharmonic_fe_system(uks,p,cp,nq=(nq or None))
            H=dev.assemble_mapped_matrix(uks,p,cpd,"biharmonic",nq=(nq or None)).to_scipy()
            if abs(Ho).max()>0:
                e=abs(H-Ho).max()/abs(Ho).max()
                assert e<=1e-10, "biharmonic: %g"%e
    except Exception as ex:
        bad+=1; print("FAIL",json.dumps(case),type(ex).__name__,str(ex)[:200],flush=True)
print(json.dumps({"cases":N,"failed":bad}))
