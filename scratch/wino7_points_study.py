import sys; sys.path.insert(0,'/root/repo/scratch')
import numpy as np
from fractions import Fraction as F
import wino_matrices as wm
def mats(points, m):
    AT,G,BT = wm.winograd(points, m); wm.check(AT,G,BT,m)
    f=lambda M: np.array([[float(x) for x in r] for r in M])
    return f(AT), f(G), f(BT)
rng=np.random.RandomState(0)
R,C,O=8,512,64
x=np.maximum(rng.randn(R,7,7,C),0)*rng.lognormal(0,1,size=(1,1,1,C))   # ReLU inputs with channel scales
w=rng.randn(3,3,C,O)/np.sqrt(9*C)
xp=np.zeros((R,9,9,C)); xp[:,1:8,1:8]=x
ref=np.zeros((R,7,7,O))
for i in range(3):
    for j in range(3): ref+=xp[:,i:i+7,j:j+7]@w[i,j]
def run(AT,G,BT, f32acc=True):
    AT32,BT32=AT.astype(np.float32),BT.astype(np.float32)
    U=np.einsum('ai,ijco,bj->abco',G,w,G).astype(np.float32)          # f64 then rounded (host transform)
    # input transform in f32
    x32=xp.astype(np.float32)
    n=BT.shape[1]
    d=x32[:,:n,:n]
    V=np.einsum('ai,rijc->rajc',BT32,d).astype(np.float32); V=np.einsum('bj,rajc->rabc',BT32,V).astype(np.float32)
    if f32acc:
        M=np.einsum('rabc,abco->rabo',V,U,dtype=np.float32)
    else:
        M=np.einsum('rabc,abco->rabo',V.astype(np.float64),U.astype(np.float64))
    M=M.astype(np.float32)
    Y=np.einsum('ia,rabo->ribo',AT32,M).astype(np.float32); Y=np.einsum('jb,ribo->rijo',AT32,Y).astype(np.float32)
    return Y
def err(Y,refpart): return np.abs(Y-refpart).max()/np.abs(ref).max()
# current: block F(4,3)+F(3,3)
A7,G7,B7=[np.array([[float(v) for v in r] for r in M]) for M in (wm.AT7,wm.G7,wm.BT7)]
Y=run(A7,G7,B7); print('F43+F33 (121) f32acc', err(Y,ref)); Y=run(A7,G7,B7,False); print('F43+F33 (121) f64acc', err(Y,ref))
# direct f32
Yd=np.zeros((R,7,7,O),np.float32)
for i in range(3):
    for j in range(3): Yd+=(xp[:,i:i+7,j:j+7].astype(np.float32)@w[i,j].astype(np.float32))
print('direct f32', err(Yd,ref))
for pts in ([0,1,-1,2,-2,F(1,2),F(-1,2),3],[0,1,-1,2,-2,F(1,2),F(-1,2),-3],[0,1,-1,2,-2,F(1,2),F(-1,2),4],[0,1,-1,2,-2,F(1,2),F(-1,2),F(1,4)],
            [0,1,-1,F(1,2),F(-1,2),2,-2,F(3,2)],[0,1,-1,F(1,2),F(-1,2),2,-2,F(-3,2)],[0,1,-1,F(1,2),F(-1,2),F(3,2),F(-3,2),2],[0,1,-1,F(1,2),F(-1,2),F(1,4),F(-1,4),2],
            [0,1,-1,F(1,2),F(-1,2),F(3,4),F(-3,4),2],[0,1,-1,F(1,2),F(-1,2),F(3,2),F(-3,2),F(1,4)]):
    AT,G,BT=mats(pts,7)
    Y=run(AT,G,BT); e1=err(Y,ref); Y=run(AT,G,BT,False); e2=err(Y,ref)
    print('F(7,3) pts',[str(p) for p in pts],'f32acc %.2e f64acc %.2e'%(e1,e2), 'max|BT| %.1f max|AT| %.1f'%(np.abs(BT).max(),np.abs(AT).max()))
