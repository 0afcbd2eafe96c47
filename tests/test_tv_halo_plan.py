"""The fused T+V kernel recomputes T on the halo pixels of a flagged 64x4 tile that some 7x7 window of a short-history pixel reaches, and
deals those cells out to its threads without holes: per staged row a 70-bit need mask from the tile's four wave masks (scalar), the
running counts, and per thread the i-th needed cell by a rank select (csrc/svgf_temporal.hip).  This replays that arithmetic on the CPU
and checks it against the definition: exactly the needed cells inside the frame / T's rows, each once."""
import random
kVR, kVW, kVH = 3, 70, 10
def plan(m, x0, W, y0, row0, row1):
    need=[]; before=[0]
    for ry in range(kVH):
        rows=0
        if ry<=6: rows|=m[0]
        if 1<=ry<=7: rows|=m[1]
        if 2<=ry<=8: rows|=m[2]
        if ry>=3: rows|=m[3]
        lo=rows; hi=0
        for d in range(1,7):
            lo|=(rows<<d)&(2**64-1); hi|=rows>>(64-d)
        if 3<=ry<7: lo&=7; hi&=~7
        ty=y0-3+ry
        if ty<row0 or ty>=row1: lo=hi=0
        if x0==0: lo&=~7
        cols_in=W-(x0-3)
        if cols_in<kVW:
            lo&= (2**64-1) if cols_in>=64 else (1<<cols_in)-1
            hi&= ((1<<(cols_in-64))-1) if cols_in>64 else 0
        hi&=0x3f
        need.append((lo,hi)); before.append(before[-1]+bin(lo).count('1')+bin(hi).count('1'))
    cells=[]
    for i in range(before[-1]):
        ry=0;j=i;lo,hi=need[0]
        for r in range(1,kVH):
            if i>=before[r]: ry=r;j=i-before[r];lo,hi=need[r]
        rx=0
        for step in (64,32,16,8,4,2,1):
            cand=rx+step
            below = bin(lo).count('1')+bin(hi&((1<<min(cand-64,31))-1)).count('1') if cand>=64 else bin(lo&((1<<cand)-1)).count('1')
            if cand<kVW and below<=j: rx=cand
        cells.append((ry,rx))
    return cells
def truth(m, x0, W, y0, row0, row1):
    out=set()
    for ry in range(kVH):
        for rx in range(kVW):
            if 3<=ry<7 and 3<=rx<67: continue
            tx=x0-3+rx; ty=y0-3+ry
            if tx<0 or tx>=W or ty<row0 or ty>=row1: continue
            ok=False
            for r in range(4):
                for c in range(64):
                    if (m[r]>>c)&1 and abs(r-(ry-3))<=3 and abs(c-(rx-3))<=3: ok=True
            if ok: out.add((ry,rx))
    return out
def test_needed_halo_cells_are_dealt_out_exactly_once():
    random.seed(1)
    for t in range(300):
        dens=random.choice([0.002,0.02,0.2,1.0])
        m=[sum((random.random()<dens)<<c for c in range(64)) for _ in range(4)]
        if not any(m): m[random.randrange(4)]|=1<<random.randrange(64)
        x0=random.choice([0,64,128,3776]); W=random.choice([3840, x0+random.randrange(1,64), x0+64, x0+66, x0+67])
        if W<=x0: W=x0+5
        y0=random.choice([0,4,100]); row0=random.choice([0,y0-2 if y0>=2 else 0]); row1=random.choice([y0+4,y0+5,y0+8,2160])
        c=plan(m,x0,W,y0,row0,row1); tr=truth(m,x0,W,y0,row0,row1)
        assert len(c)==len(set(c)) and set(c)==tr, (t, len(c), len(tr))


def test_whole_ring_shortcut_enumerates_the_ring():
    """Frames without history flag every tile and need every halo cell: the kernel then maps thread i straight to the i-th cell of the
    ring (rows 0..2, the six side columns of rows 3..6, rows 7..9) instead of searching; replayed here."""
    cells = set()
    for i in range(kVW * kVH - 256):
        if i < kVR * kVW:
            ry = i // kVW
            rx = i - ry * kVW
        elif i < kVR * kVW + 24:
            j = i - kVR * kVW
            ry = kVR + j // 6
            c = j - (j // 6) * 6
            rx = c if c < kVR else 64 + c
        else:
            j = i - kVR * kVW - 24
            ry = kVR + 4 + j // kVW
            rx = j - (j // kVW) * kVW
        assert not (3 <= ry < 7 and 3 <= rx < 67)
        cells.add((ry, rx))
    assert len(cells) == 444 and all(0 <= ry < kVH and 0 <= rx < kVW for ry, rx in cells)
