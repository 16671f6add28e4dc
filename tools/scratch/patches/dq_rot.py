SUBS = [
('''    stage(0, 0);
    stage(1, 1);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // step t has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                      // ... and the K tile's other pieces; slot (t + 2) % 3 is free again
        asm volatile("" ::: "memory");
        stage(t + 2, (t + 2) % DQ_RING);
        const char *kt = smem + (t % DQ_RING) * FTILE;
        const unsigned dsb = ds_rd + (unsigned)((t % DQ_RING) * DQ_SLAB);
        if (active && 2 * t <= sl && !KF_MUT(3, t == 0 && qblk == nxb - 1)) { // wave-uniform: key block 2 t lies at or below this slice's diagonal
            dq_step<BF, 0, 0, DB>(kt, vo, dsb, dq);
            dq_step<BF, 0, 1, DB>(kt, vo, dsb, dq);
            if (2 * t + 1 <= sl) {''',
'''    // EXPERIMENT: every workgroup walks its key tiles from its own starting point (rotation), so that the 256 concurrent streams do not sit at the same offset of their rows
    const int rot = (int)(((blockIdx.x * 2654435761u) >> 16) % (unsigned)nt);
    auto TT = [&](int t) { int x = (t < nt ? t : nt - 1) + rot; return x >= nt ? x - nt : x; };
    stage(TT(0), 0);
    stage(TT(1), 1);
    for (int t0 = 0; t0 < nt; ++t0) {
        const int t = TT(t0);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // step t has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                      // ... and the K tile's other pieces; slot (t + 2) % 3 is free again
        asm volatile("" ::: "memory");
        stage(TT(t0 + 2), (t0 + 2) % DQ_RING);
        const char *kt = smem + (t0 % DQ_RING) * FTILE;
        const unsigned dsb = ds_rd + (unsigned)((t0 % DQ_RING) * DQ_SLAB);
        if (active && 2 * t <= sl && !KF_MUT(3, t == 0 && qblk == nxb - 1)) { // wave-uniform: key block 2 t lies at or below this slice's diagonal
            dq_step<BF, 0, 0, DB>(kt, vo, dsb, dq);
            dq_step<BF, 0, 1, DB>(kt, vo, dsb, dq);
            if (2 * t + 1 <= sl) {'''),
]
