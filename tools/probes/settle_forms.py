"""Round 6: the rigid-triangle (SETTLE) solvers of vv_dev_constraints.inc in their old form (masses, three square roots + three reciprocals for the axes, the deltx refinement,
the velocity multipliers written out in the masses) and their new form (inverse masses, |Y| = |Z||X|, xb2d in closed form, the symmetric 3 x 3 system by cofactors), in numpy:
agreement of the two, bond lengths, centre of mass and momentum.   python tools/probes/settle_forms.py"""
import numpy as np
rng=np.random.default_rng(1)
def sdot(a,b): return a@b
def old_pos(m0,m1,dAB,dBB,a0,a1p,a2p,x0,x1,x2):
    b0=a1p-a0; c0=a2p-a0; iM=1/(m0+2*m1)
    com=(x0*m0+(b0+x1)*m1+(c0+x2)*m1)*iM
    A1=x0-com; B1=b0+x1-com; C1=c0+x2-com
    Z=np.cross(b0,c0); X=np.cross(A1,Z); Y=np.cross(Z,X)
    t1=X/np.linalg.norm(X); t2=Y/np.linalg.norm(Y); t3=Z/np.linalg.norm(Z)
    xb0d=t1@b0; yb0d=t2@b0; xc0d=t1@c0; yc0d=t2@c0; za1d=t3@A1
    xb1d=t1@B1; yb1d=t2@B1; zb1d=t3@B1; xc1d=t1@C1; yc1d=t2@C1; zc1d=t3@C1
    rc=0.5*dBB; rb=np.sqrt(dAB*dAB-rc*rc); ra=rb*2*m1*iM; rb-=ra
    sinphi=za1d/ra; cosphi=np.sqrt(1-sinphi**2); sinpsi=(zb1d-zc1d)/(2*rc*cosphi); cospsi=np.sqrt(1-sinpsi**2)
    ya2d=ra*cosphi; xb2d=-rc*cospsi; rcss=rc*sinpsi*sinphi; yb2d=-rb*cosphi-rcss; yc2d=-rb*cosphi+rcss
    xb2d2=xb2d*xb2d; dyb=yb2d-yc2d; dzb=zb1d-zc1d; hh2=4*xb2d2+dyb*dyb+dzb*dzb
    deltx=2*xb2d+np.sqrt(dBB*dBB+4*xb2d2-hh2); xb2d-=deltx*0.5
    return finish(t1,t2,t3,com,b0,c0,xb0d,yb0d,xc0d,yc0d,za1d,xb1d,yb1d,zb1d,xc1d,yc1d,zc1d,ya2d,xb2d,yb2d,yc2d)
def finish(t1,t2,t3,com,b0,c0,xb0d,yb0d,xc0d,yc0d,za1d,xb1d,yb1d,zb1d,xc1d,yc1d,zc1d,ya2d,xb2d,yb2d,yc2d):
    alpha=xb2d*(xb0d-xc0d)+yb0d*yb2d+yc0d*yc2d
    beta=xb2d*(yc0d-yb0d)+xb0d*yb2d+xc0d*yc2d
    gamma=xb0d*yb1d-xb1d*yb0d+xc0d*yc1d-xc1d*yc0d
    al2be2=alpha*alpha+beta*beta
    sintheta=(alpha*gamma-beta*np.sqrt(al2be2-gamma*gamma))/al2be2
    costheta=np.sqrt(1-sintheta**2)
    a3=np.array([-ya2d*sintheta, ya2d*costheta, za1d])
    b3=np.array([xb2d*costheta-yb2d*sintheta, xb2d*sintheta+yb2d*costheta, zb1d])
    c3=np.array([-xb2d*costheta-yc2d*sintheta, -xb2d*sintheta+yc2d*costheta, zc1d])
    T=np.stack([t1,t2,t3],axis=1)   # columns
    A=T@a3; B=T@b3; C=T@c3
    return com+A, com+B-b0, com+C-c0
def new_pos(iA,iB,dAB,dBB,a0,a1p,a2p,x0,x1,x2):
    b0=a1p-a0; c0=a2p-a0
    den=1/(iB+2*iA); wA=iB*den; wB=iA*den
    com=x0*wA+(b0+x1)*wB+(c0+x2)*wB
    A1=x0-com; B1=b0+x1-com; C1=c0+x2-com
    Z=np.cross(b0,c0); X=np.cross(A1,Z); Y=np.cross(Z,X)
    iax=1/np.sqrt(X@X); iaz=1/np.sqrt(Z@Z); iay=iax*iaz
    t1=X*iax; t2=Y*iay; t3=Z*iaz
    xb0d=t1@b0; yb0d=t2@b0; xc0d=t1@c0; yc0d=t2@c0; za1d=t3@A1
    xb1d=t1@B1; yb1d=t2@B1; zb1d=t3@B1; xc1d=t1@C1; yc1d=t2@C1; zc1d=t3@C1
    rc=0.5*dBB; rbt=np.sqrt(dAB*dAB-rc*rc); ra=rbt*2*wB; rb=rbt-ra
    sinphi=za1d/ra; cosphi=np.sqrt(1-sinphi**2); sinpsi=(zb1d-zc1d)/(2*rc*cosphi)
    ya2d=ra*cosphi; rcss=rc*sinpsi*sinphi; yb2d=-rb*cosphi-rcss; yc2d=-rb*cosphi+rcss
    dyb=yb2d-yc2d; dzb=zb1d-zc1d
    xb2d=-0.5*np.sqrt(dBB*dBB-(dzb*dzb+dyb*dyb))
    return finish(t1,t2,t3,com,b0,c0,xb0d,yb0d,xc0d,yc0d,za1d,xb1d,yb1d,zb1d,xc1d,yc1d,zc1d,ya2d,xb2d,yb2d,yc2d)
def old_vel(m0,m1,p0,p1,p2,v0,v1,v2):
    eAB=p1-p0; eBC=p2-p1; eCA=p0-p2
    eAB/=np.linalg.norm(eAB); eBC/=np.linalg.norm(eBC); eCA/=np.linalg.norm(eCA)
    vAB=(v1-v0)@eAB; vBC=(v2-v1)@eBC; vCA=(v0-v2)@eCA
    cA=-(eAB@eCA); cB=-(eAB@eBC); cC=-(eBC@eCA)
    s2A=1-cA*cA; s2B=1-cB*cB; s2C=1-cC*cC
    mA=m0; mB=m1; mC=m1
    mABCinv=1/(mA*mB*mC); msum=mA+mB+mC
    inner=s2B*mA*mA+2*(cA*cB*cC+1)*mA*mB+s2A*mB*mB
    denom=(s2C*mA*mB*(mA+mB)+((s2B*mA+s2A*mB)*mC+inner)*mC)*mABCinv
    kAB=s2C*mA*mA*mB*mB*mABCinv+msum; kBC=s2A*mB*mB*mC*mC*mABCinv+msum; kCA=s2B*mA*mA*mC*mC*mABCinv+msum
    oAB_CA=cB*cC*mA-cA*mB-cA*mC; oAB_BC=cA*cC*mB-cB*mC-cB*mA; oBC_CA=cA*cB*mC-cC*mB-cC*mA
    tab=(kAB*vAB+oAB_BC*vBC+oAB_CA*vCA)/denom; tbc=(oAB_BC*vAB+kBC*vBC+oBC_CA*vCA)/denom; tca=(oAB_CA*vAB+oBC_CA*vBC+kCA*vCA)/denom
    return v0+(eAB*tab-eCA*tca)/mA, v1+(eBC*tbc-eAB*tab)/mB, v2+(eCA*tca-eBC*tbc)/mC
def new_vel(iA,iB,p0,p1,p2,v0,v1,v2):
    rAB=p1-p0; rBC=p2-p1; rCA=p0-p2; iC=iB
    V=np.array([(v1-v0)@rAB,(v2-v1)@rBC,(v0-v2)@rCA])
    M=np.array([[(iA+iB)*(rAB@rAB), -iB*(rAB@rBC), -iA*(rAB@rCA)],[-iB*(rAB@rBC), (iB+iC)*(rBC@rBC), -iC*(rBC@rCA)],[-iA*(rAB@rCA), -iC*(rBC@rCA), (iC+iA)*(rCA@rCA)]])
    tab,tbc,tca=np.linalg.solve(M,V)
    return v0+(rAB*tab-rCA*tca)*iA, v1+(rBC*tbc-rAB*tab)*iB, v2+(rCA*tca-rBC*tbc)*iC
dAB=0.1; ang=np.deg2rad(109.47); dBB=2*dAB*np.sin(ang/2)
m0,m1=15.9994,1.008
for trial in range(5):
    R=np.linalg.qr(rng.normal(size=(3,3)))[0]
    a0=np.zeros(3); a1p=np.array([dAB*np.sin(ang/2), dAB*np.cos(ang/2),0]); a2p=np.array([-dAB*np.sin(ang/2), dAB*np.cos(ang/2),0])
    sh=rng.normal(size=3)
    a0,a1p,a2p=[R@p+sh for p in (a0,a1p,a2p)]
    x0,x1,x2=[rng.normal(size=3)*0.004 for _ in range(3)]
    o=old_pos(m0,m1,dAB,dBB,a0,a1p,a2p,x0,x1,x2); n=new_pos(1/m0,1/m1,dAB,dBB,a0,a1p,a2p,x0,x1,x2)
    q0,q1,q2=a0+n[0],a1p+n[1],a2p+n[2]
    print("pos diff", max(np.abs(o[i]-n[i]).max() for i in range(3)), "bonds", abs(np.linalg.norm(q1-q0)-dAB), abs(np.linalg.norm(q2-q0)-dAB), abs(np.linalg.norm(q2-q1)-dBB),
          "com", np.abs((m0*n[0]+m1*n[1]+m1*n[2])-(m0*x0+m1*x1+m1*x2)).max())
    v0,v1,v2=[rng.normal(size=3) for _ in range(3)]
    ov=old_vel(m0,m1,q0,q1,q2,v0,v1,v2); nv=new_vel(1/m0,1/m1,q0,q1,q2,v0,v1,v2)
    print("vel diff", max(np.abs(ov[i]-nv[i]).max() for i in range(3)), "rel", abs((nv[1]-nv[0])@(q1-q0)), abs((nv[2]-nv[1])@(q2-q1)), abs((nv[0]-nv[2])@(q0-q2)),
          "mom", np.abs(m0*(nv[0]-v0)+m1*(nv[1]-v1)+m1*(nv[2]-v2)).max())
