#!/usr/bin/env python3
"""Model of the tile kernel's schedule for ONE frame (32 resident workgroups, dynamic tickets, a group is counted in
the step its ticket is drawn and emitted in the next): what graded group sizes at the end of a frame would do to the
launch time.  Step = f (ticket, two barriers, the count's two dependent round trips) + items per wave x it, with the
values read off tools/wg_timeline.py and tools/stamps.py (f ~ 4 us, it ~ 3.5 us).  CPU only; DESIGN.md section 5."""
import heapq, random
def sim(n_items, workers, sizes_fn, f=4.0, it=3.5, seed=0):
    # sizes_fn(remaining_items) -> chunk size for next ticket
    rnd=random.Random(seed)
    # build ticket list
    chunks=[]; r=n_items
    while r>0:
        c=min(sizes_fn(r), r); chunks.append(c); r-=c
    nxt=0
    # each worker: (time, prev_chunk)
    pq=[(rnd.random()*2.0, w, 0) for w in range(workers)]   # staggered start
    heapq.heapify(pq); end=0; busy=0
    while pq:
        t,w,prev=heapq.heappop(pq)
        if nxt < len(chunks):
            c=chunks[nxt]; nxt+=1
            dt=f + (prev/4.0)*it*(0.9+0.2*rnd.random()) + (0 if prev else 0)
            heapq.heappush(pq,(t+dt,w,c)); busy+=dt
        else:
            if prev:
                dt=1.5+(prev/4.0)*it*(0.9+0.2*rnd.random())
                end=max(end,t+dt); busy+=dt
            else: end=max(end,t)
    return end, busy/workers, len(chunks)
N=2995; P=32
for name,fn in [("all16", lambda r:16),
                ("tail 512x8", lambda r: 16 if r>512 else 8),
                ("tail 256x8", lambda r: 16 if r>256 else 8),
                ("tail 512x8,128x4", lambda r: 16 if r>512 else (8 if r>128 else 4)),
                ("tail 1024x8,256x4", lambda r: 16 if r>1024 else (8 if r>256 else 4)),
                ("guided r/(2P)", lambda r: max(4, min(16, (r//(2*32))//4*4))),
                ("all8", lambda r:8), ("all4", lambda r:4)]:
    res=[sim(N,P,fn,seed=s) for s in range(20)]
    print(f"{name:22s} makespan {sum(r[0] for r in res)/20:7.1f} us   busy/worker {sum(r[1] for r in res)/20:7.1f}  tickets {res[0][2]}")
