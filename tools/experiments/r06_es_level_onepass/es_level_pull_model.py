import numpy as np, random
def seq_partition(key, val, first, last):
    # libstdc++ __move_median_to_first(first, first+1, mid, last-1) + __unguarded_partition(first+1, last, first)
    a, b, c = first+1, first+(last-first)//2, last-1
    ka, kb, kc = key[a], key[b], key[c]
    if ka < kb:
        if kb < kc: med = b
        elif ka < kc: med = c
        else: med = a
    elif ka < kc: med = a
    elif kb < kc: med = c
    else: med = b
    key[first], key[med] = key[med], key[first]; val[first], val[med] = val[med], val[first]
    p = key[first]
    i, j = first+1, last
    while True:
        while key[i] < p: i += 1
        j -= 1
        while p < key[j]: j -= 1
        if not (i < j): return i
        key[i], key[j] = key[j], key[i]; val[i], val[j] = val[j], val[i]
        i += 1

def tile_partition(skey, sval, first, last, T):
    n = len(skey)
    dkey, dval = skey.copy(), sval.copy()
    a, b, c = first+1, first+(last-first)//2, last-1
    ka, kb, kc = skey[a], skey[b], skey[c]
    if ka < kb:
        if kb < kc: med = b
        elif ka < kc: med = c
        else: med = a
    elif ka < kc: med = a
    elif kb < kc: med = c
    else: med = b
    km, vm, kf, vf = skey[med], sval[med], skey[first], sval[first]
    p = km
    def src(i):
        if i == first: return km, vm
        if i == med: return kf, vf
        return skey[i], sval[i]
    ta, tb = first // T, (last-1)//T
    # counts per tile
    cl, cr = {}, {}
    for t in range(ta, tb+1):
        L = R = 0
        for i in range(max(t*T, first+1), min((t+1)*T, last)):
            k = src(i)[0]
            L += k >= p; R += k <= p
        cl[t], cr[t] = L, R
    nL, nR = sum(cl.values()), sum(cr.values())
    Lp, Rl = {}, {}
    for t in range(ta, tb+1):
        pl = sum(cl[x] for x in range(ta, t)); pr = sum(cr[x] for x in range(ta, t))
        for i in range(max(t*T, first+1), min((t+1)*T, last)):
            k = src(i)[0]
            if k >= p: Lp[pl] = i; pl += 1
            if k <= p: Rl[pr] = i; pr += 1
    kmax = min(nL, nR+1)
    Rk = lambda k: Rl[nR-1-k] if k < nR else first
    cut = None
    for t in range(ta, tb+1):
        pl = sum(cl[x] for x in range(ta, t)); pr = sum(cr[x] for x in range(ta, t))
        for i in range(max(t*T, first), min((t+1)*T, last)):
            kk, vv = src(i)
            inS = first < i < last
            isL = inS and kk >= p; isR = inS and kk <= p
            if isL:
                k = pl; pl += 1
                if k < kmax:
                    bpos = Rk(k); cnd = i < bpos
                    if cnd: kk, vv = src(bpos)
                    k1 = k+1
                    cn = k1 < kmax and Lp[k1] < Rk(k1)
                    if cnd and not cn:
                        K = k1; aa = Lp[K] if K < nL else 1<<40; assert cut is None; cut = min(aa, bpos)
                    if k == 0 and not cnd:
                        assert cut is None; cut = i
            if isR:
                kr = nR-1-pr; pr += 1
                if kr < kmax and kr < nL:
                    apos = Lp[kr]
                    if apos < i:
                        kk, vv = src(apos)
            dkey[i], dval[i] = kk, vv
    return dkey, dval, cut

random.seed(1)
for it in range(3000):
    n = random.randint(20, 400)
    hi = random.choice([2,3,5,50,1000])
    key = np.array([random.randrange(hi) for _ in range(n)]); 
    if random.random()<0.3: key.sort()
    if random.random()<0.1: key = key[::-1].copy()
    val = np.arange(n)
    first = random.randint(0, n-18); last = random.randint(first+17, n)
    T = random.choice([8,16,32,64])
    k1, v1 = key.copy(), val.copy()
    cut1 = seq_partition(k1, v1, first, last)
    k2, v2, cut2 = tile_partition(key, val, first, last, T)
    assert cut1 == cut2, (it, cut1, cut2)
    assert (k1==k2).all() and (v1==v2).all(), it
print("model ok")
