"""-m gpu: every op kind of the backbone program, one at a time, through the C ABI against the
oracle's gather-GEMM-scatter (same seeded inputs, rows matched by coordinates)."""
import numpy as np
import pytest
import torch

from agile3d_amd import lib as L
from agile3d_amd.engine import Scene
from agile3d_amd.synthetic import make_scene
from gpu_util import OneOp, internal_to_oracle_rows, pack_weight
from oracle import backbone as ob

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 0], ids=["deep", "streamk"])
def conv_kernel_family(request):
    """The gathered-convolution tests run twice: with the small-level kernel (k_conv_deep: on this 6 k-voxel scene it takes every
    gathered convolution its cost model allows) and with the stream-K kernel only (a3d_conv_deep_mode)."""
    lib = L.load()
    before = lib.a3d_conv_deep_mode(request.param)
    yield request.param
    lib.a3d_conv_deep_mode(before)


@pytest.fixture(scope="module")
def world():
    coords = make_scene(6000, seed=5)["coords"]
    sc = Scene(torch.from_numpy(coords).cuda())
    lv = ob.SparseLevels(coords)
    maps = [torch.from_numpy(internal_to_oracle_rows(sc, lv, i)) for i in range(5)]
    return coords, sc, lv, maps


def _check(got, ref, what, tol=2e-4):
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    print(f"{what}: max|diff|={err:.3e} (ref scale {scale:.2f})")
    assert err <= tol * scale, (what, err)


CONV3 = [(0, 32, 32), (0, 128, 96), (1, 96, 96), (1, 32, 64), (2, 192, 128), (3, 384, 256), (4, 256, 256), (2, 64, 64)]


@pytest.mark.parametrize("level,cin,cout", CONV3)
@pytest.mark.parametrize("epi", ["plain", "bn_res_relu"])
def test_conv3(world, conv_kernel_family, level, cin, cout, epi):
    coords, sc, lv, maps = world
    g = torch.Generator().manual_seed(level * 1000 + cin + cout)
    n = sc.n[level]
    X = torch.randn(n, cin, generator=g)
    W = torch.randn(27, cin, cout, generator=g) / (cin * 14) ** 0.5
    full = epi == "bn_res_relu"
    scale = (torch.rand(cout, generator=g) + 0.5).cuda() if full else None
    shift = torch.randn(cout, generator=g).cuda() if full else None
    R = torch.randn(n, cout, generator=g) if full else None
    op = OneOp(sc, L.OP_CONV3, level, cin, cout, 27, pack_weight(W.cuda()), scale, shift, relu=full, use_res=full,
               in_pad=32 if full else 0, out_pad=32 if full else 0)
    m = maps[level]
    op.buffer(0)[:n, op.in_pad:] = X[m].cuda()
    if full:
        op.buffer(2)[:n] = R[m].cuda()
    op.buffer(1).fill_(float("nan"))
    op.run()
    ref = ob.sparse_conv(X, W, lv.kernel_map(level, 3), n)
    if full:
        ref = torch.relu(ref * scale.cpu() + shift.cpu() + R)
    out = op.buffer(1).cpu()
    _check(out[:n, op.out_pad:], ref[m], f"conv3 L{level} {cin}->{cout} {epi}")
    assert (out[n, op.out_pad:] == 0).all(), "zero row not written"
    if full:
        assert torch.isnan(out[:n, :op.out_pad]).all(), "wrote outside its column slice"


@pytest.mark.parametrize("level,c", [(0, 32), (1, 32), (2, 64), (3, 128)])
def test_down(world, conv_kernel_family, level, c):
    coords, sc, lv, maps = world
    g = torch.Generator().manual_seed(7 + level)
    n, nC = sc.n[level], sc.n[level + 1]
    X = torch.randn(n, c, generator=g)
    W = torch.randn(8, c, c, generator=g) / (c * 4) ** 0.5
    scale = (torch.rand(c, generator=g) + 0.5).cuda()
    shift = torch.randn(c, generator=g).cuda()
    op = OneOp(sc, L.OP_DOWN, level, c, c, 8, pack_weight(W.cuda()), scale, shift, relu=True)
    op.buffer(0)[:n] = X[maps[level]].cuda()
    op.run()
    ref = torch.relu(ob.sparse_conv(X, W, lv.stride_map(level), nC) * scale.cpu() + shift.cpu())
    _check(op.buffer(1).cpu()[:nC], ref[maps[level + 1]], f"down L{level}->{level + 1} {c}")


@pytest.mark.parametrize("level_in,cin,cout", [(4, 256, 256), (3, 256, 128), (2, 128, 96), (1, 96, 96)])
def test_up(world, conv_kernel_family, level_in, cin, cout):
    coords, sc, lv, maps = world
    g = torch.Generator().manual_seed(11 + level_in)
    nC, nF = sc.n[level_in], sc.n[level_in - 1]
    X = torch.randn(nC, cin, generator=g)
    W = torch.randn(8, cin, cout, generator=g) / cin ** 0.5
    shift = torch.randn(cout, generator=g).cuda()
    op = OneOp(sc, L.OP_UP, level_in, cin, cout, 8, pack_weight(W.cuda()), None, shift, relu=True, out_pad=32)
    op.buffer(0)[:nC] = X[maps[level_in]].cuda()
    op.run()
    kmap = [(rc, rf) for (rf, rc) in lv.stride_map(level_in - 1)]
    ref = torch.relu(ob.sparse_conv(X, W, kmap, nF) + shift.cpu())
    _check(op.buffer(1).cpu()[:nF, 32:], ref[maps[level_in - 1]], f"up L{level_in}->{level_in - 1} {cin}->{cout}")


@pytest.mark.parametrize("level,cin,cout", [(0, 128, 96), (2, 32, 64), (3, 384, 256), (4, 128, 256)])
def test_linear(world, level, cin, cout):
    coords, sc, lv, maps = world
    g = torch.Generator().manual_seed(13 + level)
    n = sc.n[level]
    X = torch.randn(n, cin, generator=g)
    W = torch.randn(1, cin, cout, generator=g) / cin ** 0.5
    scale = (torch.rand(cout, generator=g) + 0.5).cuda()
    shift = torch.randn(cout, generator=g).cuda()
    op = OneOp(sc, L.OP_LINEAR, level, cin, cout, 1, pack_weight(W.cuda()), scale, shift)
    op.buffer(0)[:n] = X[maps[level]].cuda()
    op.run()
    ref = (X @ W[0]) * scale.cpu() + shift.cpu()
    _check(op.buffer(1).cpu()[:n], ref[maps[level]], f"linear L{level} {cin}->{cout}")


@pytest.mark.parametrize("ks", [5, 3])
def test_stem(world, ks):
    coords, sc, lv, maps = world
    g = torch.Generator().manual_seed(17 + ks)
    n = sc.n[0]
    F3 = torch.rand(n, 3, generator=g)
    W = torch.randn(ks ** 3, 3, 32, generator=g) / 6.0
    scale = (torch.rand(32, generator=g) + 0.5).cuda()
    shift = torch.randn(32, generator=g).cuda()
    op = OneOp(sc, L.OP_STEM, 0, 3, 32, ks ** 3, W.cuda().contiguous(), scale, shift, relu=True, out_pad=96)
    op.run(F3.cuda())
    ref = torch.relu(ob.sparse_conv(F3, W, lv.kernel_map(0, ks), n) * scale.cpu() + shift.cpu())
    _check(op.buffer(1).cpu()[:n, 96:], ref[maps[0]], f"stem k{ks}")


@pytest.mark.parametrize("ks", [5, 3])
def test_stem_hash_path(ks):
    """One far-away voxel blows the bounding box up: level 0 falls back from the dense grid to the hash table."""
    coords = np.concatenate([make_scene(3000, seed=9)["coords"], np.array([[0, 80000, -3, 7]], np.int32)])
    sc = Scene(torch.from_numpy(coords).cuda())
    assert sc.grid_dims is None
    lv = ob.SparseLevels(coords)
    m = torch.from_numpy(internal_to_oracle_rows(sc, lv, 0))
    g = torch.Generator().manual_seed(23 + ks)
    n = sc.n[0]
    F3 = torch.rand(n, 3, generator=g)
    W = torch.randn(ks ** 3, 3, 32, generator=g) / 6.0
    op = OneOp(sc, L.OP_STEM, 0, 3, 32, ks ** 3, W.cuda().contiguous(), None, None, relu=False, out_pad=0)
    op.run(F3.cuda())
    ref = ob.sparse_conv(F3, W, lv.kernel_map(0, ks), n)
    _check(op.buffer(1).cpu()[:n], ref[m], f"stem k{ks} (hash)")


def test_standalone_linear_matches_matmul():
    from agile3d_amd.engine import _ptr, _stream
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    for n in (1, 100, 128, 5000):
        X = torch.randn(n, 128, generator=g).cuda()
        W = torch.randn(128, 128, generator=g).cuda() / 11.3
        b = torch.randn(128, generator=g).cuda()
        R = torch.randn(n, 128, generator=g).cuda()
        out = torch.empty(n, 128, device="cuda")
        L.check(lib.a3d_linear(_ptr(X), 128, None, 0, n, 128, 128, _ptr(pack_weight(W.unsqueeze(0))), None, _ptr(b), _ptr(R),
                               128, 0, _ptr(out), 128, None, 0, _stream()), "a3d_linear")
        ref = X.double() @ W.double() + b.double() + R.double()
        err = (out.double() - ref).abs().max().item()
        print(f"a3d_linear n={n}: max|diff| vs fp64 = {err:.3e}")
        assert err < 1e-4


@pytest.mark.parametrize("cin,cout", [(128, 128), (128, 96), (96, 128), (96, 96)])
def test_dense_linear_all_options(cin, cout):
    """k_dense: second input added on the fly, BN-style scale/shift, residual, relu, strided operands,
    ragged row counts (tile = 128 rows, wave = 32 rows, group = 16 rows)."""
    from agile3d_amd.engine import _ptr, _stream
    lib = L.load()
    g = torch.Generator().manual_seed(cin + cout)
    W = torch.randn(cin, cout, generator=g).cuda() / 9.0
    Wp = pack_weight(W.unsqueeze(0))
    scale = (torch.rand(cout, generator=g) + 0.5).cuda()
    shift = torch.randn(cout, generator=g).cuda()
    for n in (1, 15, 16, 17, 127, 128, 129, 4999, 80_001):
        ldx, ldx2, ldr, ldo = cin + 32, cin, cout + 4, cout + 64
        X = torch.randn(n, ldx, generator=g).cuda()
        X2 = torch.randn(n, ldx2, generator=g).cuda()
        R = torch.randn(n, ldr, generator=g).cuda()
        out = torch.full((n, ldo), 7.0, device="cuda")
        L.check(lib.a3d_linear(_ptr(X), ldx, _ptr(X2), ldx2, n, cin, cout, _ptr(Wp), _ptr(scale), _ptr(shift), _ptr(R),
                               ldr, 1, _ptr(out), ldo, None, 0, _stream()), "a3d_linear")
        ref = torch.relu(((X[:, :cin].double() + X2.double()) @ W.double()) * scale.double() + shift.double()
                         + R[:, :cout].double())
        err = (out[:, :cout].double() - ref).abs().max().item()
        assert err < 1e-4, (n, err)
        assert bool((out[:, cout:] == 7.0).all()), "wrote outside its columns"


def test_linear_other_shapes_use_the_conv_kernel_and_reject_in_add():
    from agile3d_amd.engine import _ptr, _stream
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    n = 3000
    X = torch.randn(n, 64, generator=g).cuda()
    W = torch.randn(64, 32, generator=g).cuda() / 8.0
    out = torch.empty(n, 32, device="cuda")
    L.check(lib.a3d_linear(_ptr(X), 64, None, 0, n, 64, 32, _ptr(pack_weight(W.unsqueeze(0))), None, None, None, 0, 0,
                           _ptr(out), 32, None, 0, _stream()), "a3d_linear")
    assert (out.double() - X.double() @ W.double()).abs().max().item() < 1e-4
    rc = lib.a3d_linear(_ptr(X), 64, _ptr(X), 64, n, 64, 32, _ptr(pack_weight(W.unsqueeze(0))), None, None, None, 0, 0,
                        _ptr(out), 32, None, 0, _stream())
    assert rc == -6      # A3D_ERR_UNSUPPORTED


@pytest.fixture(scope="module")
def small_world():
    coords = make_scene(3000, seed=14)["coords"]
    sc = Scene(torch.from_numpy(coords).cuda())
    lv = ob.SparseLevels(coords)
    maps = [torch.from_numpy(internal_to_oracle_rows(sc, lv, i)) for i in range(5)]
    return coords, sc, lv, maps


_WIDTHS_IN = [32, 64, 96, 128, 192, 256, 384]
_WIDTHS_OUT = [32, 64, 96, 128, 256]


@pytest.mark.parametrize("kind", ["conv3", "down", "up"])
@pytest.mark.parametrize("level", [0, 1, 2, 3, 4])
def test_conv_apply_every_shape_class_small_scene(small_world, conv_kernel_family, kind, level):
    """a3d_conv_apply (the training tapes' conv: forward AND input-gradient passes, so every (cin, cout) pairing occurs)
    on the 3000-voxel scene of the gradient tests, against float64 gather-GEMM-scatter: the small levels are where a
    layer is cut into shares inside tiles (hand-off), with a handful of rows per level."""
    from agile3d_amd import backward as B
    coords, sc, lv, maps = small_world
    if kind == "down" and level == 4 or kind == "up" and level == 0:
        pytest.skip("leaves the level range")
    K = 27 if kind == "conv3" else 8
    code = {"conv3": L.OP_CONV3, "down": L.OP_DOWN, "up": L.OP_UP}[kind]
    lo = level + (1 if kind == "down" else -1 if kind == "up" else 0)
    n_in, n_out = sc.n[level], sc.n[lo]
    if kind == "conv3":
        kmap = lv.kernel_map(level, 3)
    elif kind == "down":
        kmap = lv.stride_map(level)
    else:
        kmap = [(rc, rf) for (rf, rc) in lv.stride_map(level - 1)]
    worst = 0.0
    for cin in _WIDTHS_IN:
        for cout in _WIDTHS_OUT:
            g = torch.Generator().manual_seed(level * 100000 + cin * 100 + cout)
            X = torch.randn(n_in, cin, generator=g, dtype=torch.float64)
            W = torch.randn(K, cin, cout, generator=g, dtype=torch.float64) / (cin * K / 2) ** 0.5
            x = torch.zeros(n_in + 1, cin, device="cuda")
            x[:n_in] = X[maps[level]].float().cuda()
            y = B.conv_apply(sc, code, level, pack_weight(W.float().cuda()), x, cin, cout)
            ref = ob.sparse_conv(X.float().double(), W.float().double(), kmap, n_out)[maps[lo]]
            got = y[:n_out].double().cpu()
            err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            worst = max(worst, err)
            assert err <= 2e-5, (kind, level, cin, cout, err)
            assert (y[n_out] == 0).all()
    print(f"conv_apply {kind} L{level}: worst relative error {worst:.2e} over {len(_WIDTHS_IN) * len(_WIDTHS_OUT)} shapes")


@pytest.mark.parametrize("level,cin,cout,cin2", [(1, 96, 96, 128), (2, 128, 128, 192), (1, 96, 64, 64), (0, 32, 32, 64)])
def test_fused_projection_op_and_its_fallback(world, conv_kernel_family, level, cin, cout, cin2):
    """a3d_op with a fused residual projection (proj_buf / proj_cin: BasicBlock.downsample as a 28th offset of the block's
    second conv, resnet_block.py:59-61) through the C ABI against float64: the shapes the U-Net uses run the fused build;
    96 -> 64 and 32 -> 32 have no fused instantiation (stage width 96 / the LDS-resident 32-channel kernel) -- the program
    falls back to the conv + a 1x1 launch added in place instead of failing mid-forward."""
    import ctypes as C
    from agile3d_amd.engine import _ptr, _stream
    coords, sc, lv, maps = world
    lib = L.load()
    g = torch.Generator().manual_seed(level * 7919 + cin * 31 + cout + cin2)
    n = sc.n[level]
    X = torch.randn(n, cin, generator=g)
    X2 = torch.randn(n, cin2, generator=g)
    W = torch.randn(27, cin, cout, generator=g) / (cin * 14) ** 0.5
    Wp = torch.randn(1, cin2, cout, generator=g) / cin2 ** 0.5
    shift = torch.randn(cout, generator=g).cuda()
    both = torch.cat([pack_weight(W.cuda()), pack_weight(Wp.cuda())]).contiguous()
    descs = [(level, cin), (level, cout), (level, cin2 + 32)]
    bufs = (L.BufDesc * 3)(*[L.BufDesc(a, b) for a, b in descs])
    o = L.Op()
    o.kind, o.level_in, o.cin, o.cout = L.OP_CONV3, level, cin, cout
    o.in_buf, o.in_coff, o.out_buf, o.out_coff = 0, 0, 1, 0
    o.res_buf, o.res_coff, o.relu, o.kernel_volume = L.BUF_NONE, 0, 1, 27
    o.w_dev, o.scale_dev, o.shift_dev = both.data_ptr(), None, shift.data_ptr()
    o.proj_buf, o.proj_coff, o.proj_cin = 2, 32, cin2
    ops = (L.Op * 1)(o)
    nbytes = lib.a3d_program_workspace_bytes(sc.handle, bufs, 3, ops, 1)
    assert nbytes > 0, lib.a3d_last_error()
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")

    def buf(i):
        off = lib.a3d_program_buffer_offset(sc.handle, bufs, 3, i)
        lvl, ch = descs[i]
        return ws[off:off + (sc.n[lvl] + 1) * ch * 4].view(torch.float32).view(sc.n[lvl] + 1, ch)
    m = maps[level]
    buf(0)[:n] = X[m].cuda()
    buf(2)[:n, 32:] = X2[m].cuda()
    buf(1).fill_(float("nan"))
    L.check(lib.a3d_program_run(sc.handle, bufs, 3, ops, 1, None, None, 0, _ptr(ws), nbytes, _stream()), "a3d_program_run")
    torch.cuda.synchronize()
    ref = torch.relu(ob.sparse_conv(X.double(), W.double(), lv.kernel_map(level, 3), n) + X2.double() @ Wp[0].double()
                     + shift.cpu().double())
    out = buf(1).cpu()
    _check(out[:n].double(), ref[m], f"fused projection L{level} {cin}->{cout} (+{cin2})", tol=2e-5)
    assert (out[n] == 0).all(), "zero row not written"


@pytest.mark.parametrize("level,cin,cout,head", [(0, 96, 96, 128), (0, 128, 96, 128), (1, 96, 96, 128), (0, 96, 64, 128), (0, 32, 32, 64)])
def test_fused_head_op_and_its_fallback(world, level, cin, cout, head):
    """a3d_op.head_*: lin_squeeze_head (agile3d.py:43-45,179) as a second GEMM in the epilogue of the conv that produces its
    input -- the workgroup holds complete 96-column rows in exactly the operand layout of the next MFMA -- against float64,
    rows of the external output in the CALLER's order; the op's own output must be unchanged.  Level 1 and the 64- / 32-column
    shapes have no fused build: the library runs the head as its own launch (level 1: not a level-0 op -> refused)."""
    from agile3d_amd.engine import _ptr, _stream
    coords, sc, lv, maps = world
    lib = L.load()
    g = torch.Generator().manual_seed(level * 977 + cin * 13 + cout + head)
    n = sc.n[level]
    X = torch.randn(n, cin, generator=g)
    W = torch.randn(27, cin, cout, generator=g) / (cin * 14) ** 0.5
    Wh = torch.randn(1, cout, head, generator=g) / cout ** 0.5
    bias = torch.randn(head, generator=g).cuda()
    scale, shift = (torch.rand(cout, generator=g) + 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    R = torch.randn(n, cout, generator=g)
    wp, hp = pack_weight(W.cuda()), pack_weight(Wh.cuda())
    descs = [(level, cin), (level, cout), (level, cout)]
    bufs = (L.BufDesc * 3)(*[L.BufDesc(a, b) for a, b in descs])
    o = L.Op()
    o.kind, o.level_in, o.cin, o.cout = L.OP_CONV3, level, cin, cout
    o.in_buf, o.in_coff, o.out_buf, o.out_coff = 0, 0, 1, 0
    o.res_buf, o.res_coff, o.relu, o.kernel_volume = 2, 0, 1, 27
    o.w_dev, o.scale_dev, o.shift_dev = wp.data_ptr(), scale.data_ptr(), shift.data_ptr()
    o.head_w_dev, o.head_bias_dev, o.head_cout = hp.data_ptr(), bias.data_ptr(), head
    ops = (L.Op * 1)(o)
    nbytes = lib.a3d_program_workspace_bytes(sc.handle, bufs, 3, ops, 1)
    assert nbytes > 0, lib.a3d_last_error()
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")

    def buf(i):
        off = lib.a3d_program_buffer_offset(sc.handle, bufs, 3, i)
        lvl, ch = descs[i]
        return ws[off:off + (sc.n[lvl] + 1) * ch * 4].view(torch.float32).view(sc.n[lvl] + 1, ch)
    m = maps[level]
    buf(0)[:n] = X[m].cuda()
    buf(2)[:n] = R[m].cuda()
    ext = torch.full((sc.n[0], head + 32), float("nan"), device="cuda")
    rc = lib.a3d_program_run(sc.handle, bufs, 3, ops, 1, None, _ptr(ext), head + 32, _ptr(ws), nbytes, _stream())
    if level != 0:
        assert rc != 0                         # a head writes the level-0 external output
        return
    L.check(rc, "a3d_program_run")
    torch.cuda.synchronize()
    y_ref = torch.relu(ob.sparse_conv(X.double(), W.double(), lv.kernel_map(level, 3), n) * scale.cpu().double()
                       + shift.cpu().double() + R.double())
    h_ref = y_ref @ Wh[0].double() + bias.cpu().double()
    _check(buf(1).cpu()[:n].double(), y_ref[m], f"conv with head L{level} {cin}->{cout}: own output", tol=2e-5)
    # the scene was built from `coords` in the oracle's (= caller's) row order: ext rows are caller rows
    got = ext.cpu()
    _check(got[:, :head].double(), h_ref, f"fused head {cout}->{head}", tol=2e-5)
    assert torch.isnan(got[:, head:]).all(), "wrote outside the head's columns"


def test_emulated_fp32_build_keeps_parity():
    """A3D_CONV_EMU=2 (every gathered conv kernel forms its fp32 products from six bf16-MFMA terms; read once per process, so
    it runs in its own interpreter): the conv tests that reach those kernels and the end-to-end smoke comparison with the
    oracle, at their unchanged tolerances."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, A3D_CONV_EMU="2")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_conv.py"), "-x", "-q", "-p",
                          "no:cacheprovider", "-k", "test_conv3 or test_up or test_down or every_shape_class"], env=env, capture_output=True,
                         text=True, timeout=1200, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:]
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], env=env, capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0 and "smoke ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    print(out.stdout.strip().splitlines()[-2])


def test_emulated_fp32_products_error_bound(tmp_path):
    """The opt-in emulated-fp32 build (A3D_CONV_EMU=2: an fp32 product = six bf16-MFMA terms of the three-plane operand split,
    fp32 accumulation) BOUNDED on adversarial inputs, shape class by shape class (tests/emu_cases.py: wide dynamic range,
    engineered cancellation, same-sign sums, tiny magnitudes) against a float64 evaluation of the same sums.  Errors are
    normalised by sum|x||w| of the output element (the forward-error measure of a dot product):
    (i)   a-priori: |emulated - sum| <= (2^-20 + 6 (K cin / 32) 2^-24) sum|x||w| -- the three dropped cross terms of a product
          are below 2^-21, 2^-21 and 2^-28 of it, each of the 6 K cin / 32 accumulating MFMAs rounds once;
    (ii)  measured, per shape class and family: below 2^-19, and at most twice the exact-fp32 MFMA chain's error on the same
          inputs (a lone dominant product is where the emulation is the worse of the two: its truncated cross terms against
          one fp32 rounding; measured up to 1.6x there);
    (iii) measured, per family: the worst case over the shape classes is no worse than the exact chain's worst case (x 1.25);
    (iv)  OUTSIDE the domain (|x| < 2^-100: the low planes are subnormal in bf16 and the matrix cores drop them) the result is
          what two planes give -- below 2^-15 of sum|x||w| -- while the exact chain keeps its 2^-21: reported, bounded, and the
          reason DESIGN.md states the domain."""
    import os
    import subprocess
    import sys
    import emu_cases as ec
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for mode in ("0", "2"):
        path = str(tmp_path / f"emu{mode}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "emu_cases.py"), path], env=dict(os.environ, A3D_CONV_EMU=mode),
                           capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[mode] = np.load(path)
    sc, lv, maps = ec.world()
    worst = {}
    for si, shape in enumerate(ec.SHAPES):
        kind, level, cin, cout = shape
        n_in, n_out, _, K, kmap = ec.geometry(shape, sc, lv)
        for fi, family in enumerate(ec.FAMILIES):
            X, W = ec.inputs(family, n_in, cin, cout, K, 1000 * si + fi)
            ref = ob.sparse_conv(X.double(), W.double(), kmap, n_out).numpy()
            mag = ob.sparse_conv(X.double().abs(), W.double().abs(), kmap, n_out).numpy()
            name = ec.name_of(shape, family)
            exact, emu = outs["0"][name].astype(np.float64), outs["2"][name].astype(np.float64)
            assert np.isfinite(exact).all() and np.isfinite(emu).all(), name
            live = mag > 0
            e_exact = float((np.abs(exact - ref)[live] / mag[live]).max())
            e_emu = float((np.abs(emu - ref)[live] / mag[live]).max())
            apriori = 2.0 ** -20 + 6 * (K * cin / 32) * 2.0 ** -24
            print(f"{name:34s} exact {e_exact:.3e}  emulated {e_emu:.3e}  a-priori {apriori:.3e}  (of sum|x||w|)")
            worst.setdefault(family, []).append((name, e_exact, e_emu))
            if family == "subnormal":
                assert e_emu <= 2.0 ** -15, (name, e_emu)
                continue
            assert e_emu <= apriori, (name, e_emu, apriori)
            assert e_emu <= 2.0 ** -19, (name, e_emu)
            assert e_emu <= 2.0 * e_exact + 2.0 ** -24, (name, e_emu, e_exact)
    for family, rows in worst.items():
        w_exact, w_emu = max(r[1] for r in rows), max(r[2] for r in rows)
        print(f"{family}: worst exact {w_exact:.3e}, worst emulated {w_emu:.3e}")
        if family != "subnormal":
            assert w_emu <= 1.25 * w_exact, (family, w_emu, w_exact)
