"""Debug probe 5 (needs debug_scaffolding.patch: jg_debug_xl_buf): which intermediate of the folded XLM-RoBERTa pass differs first when a
two-lane run differs?  One encoder layer, so every intermediate survives the pass:  qkv <- qkv GEMM (implicit-LN consumer), att <- attention,
stats <- out-proj GEMM (producer) + ln_stats, hid <- ff1 GEMM (consumer), xh / xl / part <- ff2 GEMM (producer).
JG_XL_DUMP=1 python tools/experiments/xlmr_race/xl_dump_probe.py [B L runs layers]"""
import sys, os, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
RUNS = int(sys.argv[3]) if len(sys.argv) > 3 else 300
LAYERS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
NAMES = ["xh", "xl", "part", "stats", "qkv", "att", "hid"]
ORDER = ["qkv", "att", "stats", "hid", "part", "xh", "xl"]          # dataflow order
ids, mask = synth.xlmr_inputs(3, B, L)
eng = Engine(0)
eng.set_option("xlmr_lanes", 2)
for kv in os.environ.get("OPTS", "").split(","):
    if kv:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
SD = synth.xlmr_state_dict(layers=LAYERS)
xl = XLMRoberta(engine=eng).load_state_dict(SD)


def host_terms(rows, c):
    """For rows of the SECOND part (lane 1) and qkv column c, in float64: the accumulator term acc = W' . x, the row statistics (mean, rstd)
    of the embedding sum x, the folded bias b' = b + W . beta and the column sum c1 = sum W' (W' = W * gamma): v = acc * rstd - c1 * mean * rstd + b'."""
    g = lambda k: np.asarray(SD[k], np.float64)
    i64 = ids.astype(np.int64)
    nonpad = (i64 != 1).astype(np.int64)
    pos = np.cumsum(nonpad, 1) * nonpad + 1
    half = B // 2
    bb, tt = half + rows // L, rows % L
    x = g("embeddings.word_embeddings.weight")[i64[bb, tt]] + g("embeddings.token_type_embeddings.weight")[0] + g("embeddings.position_embeddings.weight")[pos[bb, tt]]
    m = x.mean(1)
    rs = 1.0 / np.sqrt(x.var(1) + 1e-5)
    nm = ["query", "key", "value"][c // 768]
    w = g(f"encoder.layer.0.attention.self.{nm}.weight")[c % 768]
    bias = g(f"encoder.layer.0.attention.self.{nm}.bias")[c % 768]
    gam, bet = g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias")
    wf = w * gam
    return x @ wf, m, rs, bias + w @ bet, wf.sum()

ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
fn = eng.lib.jg_debug_xl_buf
fn.restype = ctypes.c_long
fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]


def dump():
    torch.cuda.synchronize()
    out = {}
    for lane in range(2):
        for w, nm in enumerate(NAMES):
            buf = np.empty(64 << 20, np.uint8)
            n = fn(lane, w, buf.ctypes.data, buf.nbytes)
            assert n > 0, (lane, nm, n)
            out[lane, nm] = buf[:n].copy()
    return out


eng.set_option("xlmr_lanes", 1)
one = xl(ids_d, attention_mask=mask_d).last_hidden_state.clone()
eng.set_option("xlmr_lanes", 2)
while True:
    base = xl(ids_d, attention_mask=mask_d).last_hidden_state.clone()
    ref = dump()
    if torch.equal(base, one):
        break
accbuf = accref = None
if os.environ.get("ACC") and LAYERS == 1:
    eng.lib.jg_debug_xl_ptr.restype = ctypes.c_void_p
    eng.lib.jg_debug_xl_ptr.argtypes = [ctypes.c_int, ctypes.c_int]
    eng.lib.jg_debug_set_acc.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    accbuf = torch.zeros((B - B // 2) * L, 2304, device="cuda")
    assert eng.lib.jg_debug_set_acc(accbuf.data_ptr(), eng.lib.jg_debug_xl_ptr(1, 4)) == 0
    while True:
        o2 = xl(ids_d, attention_mask=mask_d).last_hidden_state
        accref = accbuf.clone()
        if torch.equal(o2, base):
            break
eng.set_option("ws_poison", 1)
nbad = 0
for it in range(RUNS):
    if os.environ.get("TILES"):
        eng.set_option("gemm_tile", it & 3)
    out = xl(ids_d, attention_mask=mask_d).last_hidden_state
    if torch.equal(out, base):
        continue
    nbad += 1
    cur = dump()
    line = []
    for lane in range(2):
        for nm in ORDER:
            a, b = cur[lane, nm], ref[lane, nm]
            if not np.array_equal(a, b):
                esz = 4 if nm in ("part", "stats") else 2
                width = {"qkv": 2304, "hid": 3072, "part": 24, "stats": 2}.get(nm, 768)
                ne = np.flatnonzero((a.view(np.uint16 if esz == 2 else np.uint32) != b.view(np.uint16 if esz == 2 else np.uint32)))
                rows = np.unique(ne // width)
                cols = np.unique(ne % width)
                line.append(f"lane{lane}.{nm}: {len(ne)} elems, rows {rows.min()}..{rows.max()} ({len(rows)}), cols {cols.min()}..{cols.max()} ({len(cols)})")
    if LAYERS == 1 and not np.array_equal(cur[1, "qkv"], ref[1, "qkv"]):
        a16, b16 = cur[1, "qkv"].view(np.float16).reshape(-1, 2304), ref[1, "qkv"].view(np.float16).reshape(-1, 2304)
        rr, cc = np.nonzero(a16.view(np.uint16) != b16.view(np.uint16))
        for c in np.unique(cc)[:3]:
            r = rr[cc == c]
            r0 = (r.min() // 16) * 16
            blk = slice(r0, r0 + 16)
            acc_h, m_h, rs_h, b_h, c1_h = host_terms(np.arange(r0, r0 + 16), int(c))
            want_h = acc_h * rs_h - c1_h * m_h * rs_h + b_h
            h0 = b_h - c1_h * m_h * rs_h                         # the accumulator term missing
            h1 = acc_h * m_h + h0                                # the accumulator scaled by the mean instead of rstd
            h2 = acc_h * rs_h + b_h                              # the column-sum term missing
            gotf = a16[blk, c].astype(np.float64)
            err = lambda pred: float(np.abs(pred - gotf).max())
            print(f"  HYPOTHESES col {c} rows {r0}..: max |host want - gpu want| {float(np.abs(want_h - b16[blk, c].astype(np.float64)).max()):.4f}; "
                  f"max |prediction - got|: no accumulator term {err(h0):.4f}, accumulator x mean {err(h1):.4f}, no column-sum term {err(h2):.4f}, zero {err(0 * h0):.4f}")
            jj = (r0 % 32) // 16
            prev = b16[r0 - 16:r0, c] if r0 >= 16 else None
            print(f"  EVENT col {c} (mod 64: {c % 64}) rows {r0}.. j={jj} (128x128 tile) got==want[rows-16]: {prev is not None and np.array_equal(prev.view(np.uint16), a16[blk, c].view(np.uint16))}"
                  f" n_equal_to_prev {int((prev.view(np.uint16) == a16[blk, c].view(np.uint16)).sum()) if prev is not None else -1} n_wrong {int((a16[blk, c].view(np.uint16) != b16[blk, c].view(np.uint16)).sum())}")
            print(f"  qkv col {c} rows {r0}..{r0 + 15}:\n    got  {a16[blk, c]}\n    want {b16[blk, c]}\n    want rows-16 {b16[max(r0 - 16, 0):max(r0 - 16, 0) + 16, c]}"
                  f"\n    want rows+16 {b16[r0 + 16:r0 + 32, c]}\n    got/want {(a16[blk, c].astype(np.float32) / b16[blk, c].astype(np.float32))}")
            # is `got` some other column of the same rows?
            if accbuf is not None:
                print(f"    raw accumulators got  {accbuf[r0:r0 + 16, c].cpu().numpy()}\n    raw accumulators want {accref[r0:r0 + 16, c].cpu().numpy()}")
                na = int((accbuf != accref).sum())
                print(f"    accumulator elements that differ in the whole GEMM: {na}")
            hits = [int(k) for k in range(2304) if np.array_equal(b16[blk, k].view(np.uint16), a16[blk, c].view(np.uint16))]
            print(f"    columns of the reference equal to `got` on these rows: {hits}")
    d = (out - base).abs().amax(-1).cpu().numpy()
    print(f"run {it}: output differs in sequences {sorted(set(int(b) for b, _ in np.argwhere(d > 0)))}; " + ("; ".join(line) if line else "NO intermediate differs"), flush=True)
print(f"B {B} L {L} layers {LAYERS}: {nbad} of {RUNS} runs differ")
try:
    fnc = eng.lib.jg_canary_read
    cb = (ctypes.c_uint * 4096)()
    fnc(cb, 0)
    print(f"dual accumulation: accumulator quads that differ between the two sets {cb[0]}, tiles checked {cb[1]}; epilogue check records {cb[2]}")
    f = lambda u: float(np.array([u], np.uint32).view(np.float32)[0])
    h = lambda u: np.array([u], np.uint32).view(np.float16)
    for k in range(min(cb[2], 60)):
        r = cb[8 + 16 * k: 24 + 16 * k]
        print(f"  wg {r[0]} wave {r[1] >> 16} lane {r[1] >> 8 & 255} i {r[1] >> 4 & 15} j {r[1] & 15} flags(s2 v!=v2, s3 readback!=written, sg second gsc read differs) {r[2]:03b} "
              f"readback {h(r[3])} {h(r[4])} written {h(r[5])} {h(r[6])} v.x {f(r[7]):.4f} v2.x {f(r[8]):.4f} v.z {f(r[9]):.4f} v2.z {f(r[10]):.4f} "
              f"gsc.x {f(r[11]):.5f} again {f(r[12]):.5f} gsc.z {f(r[13]):.5f} again {f(r[14]):.5f} acc.x {f(r[15]):.4f}")
except AttributeError:
    pass
