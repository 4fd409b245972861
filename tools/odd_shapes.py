"""Training steps at shapes that are not multiples of any tile (ragged lengths, odd batch): losses finite, and the step with the
side-stream machinery (CTC branch, mask prefetch, split-K) agrees with the plain serial step on the same seed."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, asr_amd, bench
    dev = torch.device("cuda", 0)
    out = {}
    for (B, T, U) in [(5, 777, 33), (3, 2100, 90), (17, 130, 7), (1, 64, 1)]:
        torch.manual_seed(B * 1000 + T)
        model = bench.build_model(asr_amd, dev, 0.1, True)
        asr_amd.manual_seed(7)
        x = torch.randn(B, T, 80, device=dev)
        lens = torch.randint(T // 2, T + 1, (B,), device=dev); lens[0] = T
        tg = torch.randint(4, 4233, (B, U), device=dev)
        tl = torch.randint(max(1, U // 2), U + 1, (B,)); tl[0] = U
        for b in range(B): tg[b, int(tl[b]):] = 0
        tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
        if os.environ.get("ODD_SERIAL") == "1":      # everything on one stream, masks hashed in line
            from asr_amd import modules
            tr.overlap_ctc = tr.wgrad_stream = False
            modules._MASK_PREFETCH = False
        vals = []
        for _ in range(3):
            o = tr.step(x, lens, tg, max_target_len=U)
            vals.append([float(v) for v in o[:2]])
        torch.cuda.synchronize()
        out["%dx%dx%d" % (B, T, U)] = vals
    print("RESULT " + json.dumps(out))
else:
    res = []
    for env in ({}, {"ODD_SERIAL": "1"}):
        e = dict(os.environ); e.update(env)
        p = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(p.stdout[-2000:], p.stderr[-3000:]); sys.exit(1)
        res.append(json.loads(line[0][7:]))
    import math
    for k in res[0]:
        a, b = res[0][k], res[1][k]
        ok = all(math.isfinite(v) for s in a + b for v in s)
        rel = max(abs(x - y) / max(abs(y), 1e-6) for s, t in zip(a, b) for x, y in zip(s, t))
        print(k, "finite" if ok else "NON-FINITE", "max rel diff overlapped vs serial %.2e" % rel, a[-1], b[-1])
