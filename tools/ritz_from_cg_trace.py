"""Ritz values (estimates of the extreme eigenvalues of the preconditioned reduced system) from the Lanczos coefficients
the C++ oracle's PCG prints with ORC_CG_TRACE=1:

    ORC_CG_TRACE=1 [ORC_DEFLATE=1] python -c "... oracle.cpu.gp_solve(...)" 2> trace.txt
    python tools/ritz_from_cg_trace.py trace.txt <solve index> [<solve index> ...]

Used for DESIGN.md section 7 item 2: at configs[2], LM iteration 15, the plain solve sees 4 eigenvalues below 4e-4 and the
rest in [0.25, 1.75]; with the four gauge modes deflated the small ones are gone (46 -> 19 iterations)."""
import sys, re, numpy as np
# parse a trace: solves separated by it==0
solves=[]; cur=None
for ln in open(sys.argv[1]):
    if ln.startswith("[orc cg]"):
        t=ln.split(); it=int(t[3]); a=float(t[5]); rz=float(t[7])
        if it==0:
            cur=[]; solves.append(cur)
        cur.append((a,rz))
for si in map(int, sys.argv[2:]):
    s=solves[si]; m=len(s)
    al=np.array([a for a,_ in s]); rz=np.array([r for _,r in s])
    be=rz[1:]/rz[:-1]
    T=np.zeros((m,m))
    for j in range(m):
        T[j,j]=1/al[j]+(be[j-1]/al[j-1] if j>0 else 0)
        if j+1<m:
            T[j,j+1]=T[j+1,j]=np.sqrt(be[j])/al[j]
    w=np.linalg.eigvalsh(T)
    print(f"solve {si}: {m} iterations; Ritz values: smallest {' '.join(f'{v:.2e}' for v in w[:8])} ... largest {' '.join(f'{v:.2f}' for v in w[-3:])}")
