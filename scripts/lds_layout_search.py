"""Search LDS strides for the contraction buffers that make every access pattern of the apply kernel
bank-conflict free (ds_read_b64: 32-lane groups over 32 double-banks; ds_write_b64: 16-lane groups
over 16 double-banks).  Buffer A[qx][j][k], B[qx][qy][k]; lanes t = ta + Q1*tb, EPW elements/wave."""
import itertools, sys

def cost(addr_fn, Q1, EPW, estride, na, nb):
    # addr_fn(ta,tb) -> double index within element, lanes with ta<na,tb<nb active
    T = Q1*Q1
    lanes = []
    for lane in range(64):
        sub, t = divmod(lane, T)
        if sub >= EPW: lanes.append(None); continue
        ta, tb = t % Q1, t // Q1
        lanes.append(sub*estride + addr_fn(ta,tb) if (ta<na and tb<nb) else None)
    def grp(groups, banks):
        cyc = 0
        for g in groups:
            per = {}
            for l in g:
                a = lanes[l]
                if a is None: continue
                per.setdefault(a % banks, set()).add(a)
            cyc += max([len(v) for v in per.values()] + [1])
        return cyc
    rd = grp([range(0,32), range(32,64)], 32)
    wr = grp([range(16*i,16*i+16) for i in range(4)], 16)
    return rd, wr

def search(P1, Q1):
    NC = P1+1; T=Q1*Q1; EPW = 64//T
    best = None
    for Sj in range(NC, NC+3):
      for Sq in range(Sj*NC, Sj*NC+9):
        # A patterns: P1: (ta->j, tb->k) qx const ; P2: (ta->qx, tb->k) j const
        for Ty in range(NC, NC+3):
          for Tq in range(Ty*Q1, Ty*Q1+9):
            AF, BF = Sq*Q1, Tq*Q1
            elem = 2*AF + 3*BF
            for epad in range(0, 33, 1):
                E = elem + epad
                tot = 0
                # A
                r,w = cost(lambda ta,tb: ta*Sj+tb, Q1, EPW, E, NC, NC); tot += (r-2) + (w-4)
                r,w = cost(lambda ta,tb: ta*Sq+tb, Q1, EPW, E, Q1, NC); tot += (r-2) + (w-4)
                r,w = cost(lambda ta,tb: ta*Tq+tb, Q1, EPW, E, Q1, NC); tot += (r-2) + (w-4)
                r,w = cost(lambda ta,tb: ta*Tq+tb*Ty, Q1, EPW, E, Q1, Q1); tot += (r-2) + (w-4)
                key = (tot, E)
                if best is None or key < best[0]:
                    best = (key, dict(Sj=Sj,Sq=Sq,Ty=Ty,Tq=Tq,E=E,elem=elem))
    return best

for P1,Q1 in [(3,4),(2,4),(1,4),(4,5),(2,3),(1,2)]:
    b = search(P1,Q1)
    # baseline
    NC=P1+1
    print(P1,Q1,'best extra cycles',b[0][0], b[1])
