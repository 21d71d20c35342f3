"""Experiments build only (make -C lettuce_amd/csrc EXPERIMENTS=1; LT_ENGINE_LIBRARY=<that library>): the two-step kernel on
populations laid out in chunks of 8 planes, [chunk][q][plane in chunk][a1][a0] (lbm2_kernel<..., CHUNK = 8>, shift policy 6),
against the product layout [q][a2][a1][a0] with the resident pad: bit-identity of one double step (the chunked buffer is
filled / read back with torch indexing) and ms per update at 256^3 / 384^3 / 512^3.  DESIGN.md section 4, "Larger grids".
usage: chunked_layout_probe.py [edge ...]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lettuce_amd._native import Plan, experiments_built

assert experiments_built(), "needs the experiments build of the library"
dev = torch.device("cuda:0")
C, PAD, Q = 8, 8256, 19
for n in [int(a) for a in sys.argv[1:]] or [256, 384, 512]:
    nodes, plane = n ** 3, n * n
    blk = C * plane + PAD
    nchunks = n // C
    # product layout (resident pad)
    ref = Plan("D3Q19", torch.float32, "bgk", [n] * 3, [], device=dev)
    ref.set_population_stride(nodes + 32832 + 64)
    a = ref.empty_populations()
    for q in range(Q):
        a[q].fill_(0.02 + 0.001 * q); a[q] += 0.001 * torch.rand([n] * 3, device=dev)
    b = ref.empty_populations()
    # chunked layout: a flat buffer of nchunks * Q blocks; the plan only has to accept tensors of that size
    ch = Plan("D3Q19", torch.float32, "bgk", [n] * 3, [], device=dev)
    ch.set_population_stride(nchunks * blk)
    ch.set_shift_policy(6)
    ca, cb = ch.empty_populations(), ch.empty_populations()
    flat_a = ca.as_strided([nchunks, Q, blk], [Q * blk, blk, 1])
    flat_b = cb.as_strided([nchunks, Q, blk], [Q * blk, blk, 1])
    flat_a.zero_()
    for c in range(nchunks):                       # reference layout: a2 is the FIRST grid axis of [q, a2, a1, a0]
        flat_a[c, :, :C * plane] = a[:, c * C:(c + 1) * C].reshape(Q, C * plane)
    ref.set_two_step(1, 128 if n % 128 == 0 else 64)
    ch.set_two_step(1, 128 if n % 128 == 0 else 64)
    ref.stream_collide_twice(a, b, 0.6)
    ch.stream_collide_twice(ca, cb, 0.6)
    torch.cuda.synchronize()
    same = all(bool(torch.equal(flat_b[c, :, :C * plane].reshape(Q, C, n, n), b[:, c * C:(c + 1) * C])) for c in range(nchunks))
    row = {"grid": [n] * 3, "bit_identical_double_step": same, "kernel_chunked": ch.kernel_name()}
    for label, plan, x, y in (("product_layout", ref, a, b), ("chunked_layout", ch, ca, cb)):
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20 if n <= 384 else 8
            e0.record()
            for _ in range(reps):
                plan.stream_collide_twice(x, y, 0.6); plan.stream_collide_twice(y, x, 0.6)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (4 * reps))
        row[label + "_ms_per_update"] = round(best, 5)
        row[label + "_ps_per_node_and_update"] = round(best * 1e9 / nodes, 2)
    print(json.dumps(row), flush=True)
    del a, b, ca, cb, flat_a, flat_b, ref, ch
    torch.cuda.empty_cache()
