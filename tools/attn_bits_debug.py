import sys; sys.path.insert(0, "/root/repo")
import torch
from cldrd_amd import hip_ops as ops
DEV = "cuda"
import sys as _s
nseq, L, H, p = (70, 64, 8, 0.1) if len(_s.argv) > 1 else (48, 128, 12, 0.1)
d, T = H * 64, nseq * L
for dt in (torch.bfloat16, torch.float16):
    g = torch.Generator(device=DEV).manual_seed(7)
    qkv = torch.randn(T, 3 * d, device=DEV, generator=g).to(dt)
    dctx = torch.randn(T, d, device=DEV, generator=g).to(dt)
    mask = torch.ones(nseq, L, dtype=torch.int64, device=DEV)
    ctx = torch.empty(T, d, dtype=dt, device=DEV)
    lse = torch.empty(nseq, H, L, dtype=torch.float32, device=DEV)
    bits = ops.attention_drop_bits(nseq, L, H, p, DEV)
    bits.fill_(0)
    ops.attention_fwd(qkv, mask, ctx, lse, nseq, L, H, dropout_p=p, seed=99, drop_bits=bits)
    torch.cuda.synchronize()
    print(dt, "bits nonzero words", int((bits != 0).sum()), "of", bits.numel(), "popcount frac", float(sum(bin(int(x) & 0xFFFFFFFF).count("1") for x in bits[:2000].tolist())) / (2000 * 32))
    outs = []
    for two_role, b in ((0, None), (1, None), (1, bits)):
        ops.set_tuning("attn_bwd2", two_role)
        dq = torch.full((T, 3 * d), float("nan"), dtype=dt, device=DEV)
        ops.attention_bwd(qkv, mask, ctx, dctx, lse, dq, nseq, L, H, dropout_p=p, seed=99, drop_bits=b)
        ops.set_tuning("attn_bwd2", 1)
        outs.append(dq.float())
    for i, j in ((0, 1), (1, 2)):
        diff = (outs[i] - outs[j]).abs()
        print(dt, i, j, "n diff", int((diff > 0).sum()), "max", float(diff.max()), "cols with diff (q,k,v thirds)", [int((diff[:, k * d:(k + 1) * d] > 0).sum()) for k in range(3)])
    if dt == torch.float16:
        diff = (outs[1] - outs[2])[:, 2 * d:].abs() > 0
        rows, cols = diff.nonzero(as_tuple=True)
        print("distinct rows", rows.unique().numel(), "seq hist", torch.bincount(rows // L, minlength=nseq).tolist())
        print("key%32 hist", torch.bincount(rows % 32, minlength=32).tolist())
        print("head hist", torch.bincount(cols // 64, minlength=H).tolist())
        print("d hist", torch.bincount(cols % 64, minlength=64).tolist())
        v = outs[1][:, 2 * d:][diff]
        print("|dV| at diffs: min", float(v.abs().min()), "max", float(v.abs().max()), "mean", float(v.abs().mean()))
        d1, d2 = outs[1][:, 2 * d:], outs[2][:, 2 * d:]
        dd = (d1 - d2).abs()
        i = int(dd.argmax()); r, c = divmod(i, d)
        print("largest diff", float(dd.max()), "at", r, c, "values", float(d1[r, c]), float(d2[r, c]), "row diffs", dd[r, (c // 64) * 64:(c // 64) * 64 + 64].tolist())
