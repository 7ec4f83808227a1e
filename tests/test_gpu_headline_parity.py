"""BASELINE.json configs[1] at its full size: 100 000 Gaussians on the synthetic human surface, 1024 x 1024, forward AND
backward through the C-ABI against the CPU oracle (all host cores), one view and the 4-view launch set.

Bars: integer buffers bit-exact, images within 1e-4 (test_gpu_raster_parity._check_forward_scene); every gradient
tensor (the six parameter gradients + means2D) within
  * MAX_TOL  of the largest |gradient| of that tensor (the bar of the small-size tests), both pipelines end to end, and
  * REL_TOL  relative, element by element, on every entry whose magnitude exceeds FLOOR_FRAC x the tensor maximum
    (a Gaussian whose gradient is 1e-3 of the maximum can no longer be 200 % wrong and pass), with the oracle's
    backward given the SAME alpha image as the HIP backward.  Why: the fork's backward derives every T_j of a pixel
    from T_final := 1 - alpha_out; at a nearly opaque pixel (T_final ~ 1e-4) the 1e-7 rounding difference between two
    forwards' alpha is a percent-level difference in all of that pixel's T_j, so element-wise agreement of two
    end-to-end pipelines is not defined there (measured: 13 of 300 000 colour-gradient entries, all at one pixel).
    Handing both backwards the same forward output compares the backward itself.
    Gaussians that walk through a PROVEN knife-edge pixel (a threshold test of the blend within 2e-5 of flipping in
    the oracle: oracle.knife_edge_gaussians) are left out of the element-wise bar only — v_exp_f32 and libm expf may
    decide such a pixel differently, which moves that Gaussian's gradient by one whole pixel contribution; their
    count is printed and bounded.  Gaussians that blend at the SAME pixel see a second-order change: behind the subject
    their transmittance changes by the factor (1 - alpha) = 0.4 %, in front of it the colour accumulated behind them
    (the reverse-order backward's accum_rec) gains or loses the subject's term; that matters only where a gradient
    entry is a small sum of cancelling terms.  Those rows (oracle: sharing=True; a few per cent of the Gaussians) get the
    looser REL_TOL_DOWNSTREAM, and at most MAX_LOOSE_ENTRIES of their entries per tensor may actually exceed REL_TOL
    (measured: none, once one).
The achieved errors are printed (pytest -s) and written to gpurun_out/parity_headline.json when that directory exists.
Reference call sites: gaussian_renderer/__init__.py:85-93, threestudio/systems/GaussianIP.py:452-457."""
import json
import os

import numpy as np
import pytest
import torch

import scenes
from test_gpu_raster_parity import _assert_images, _check_forward_scene, _dev, _oracle_forward, _settings

pytestmark = pytest.mark.gpu

H = W = 1024
P = 100000
MAX_TOL = 2e-3
REL_TOL = 1e-2
FLOOR_FRAC = 1e-3
REL_TOL_DOWNSTREAM = 5e-2
MAX_LOOSE_ENTRIES = 4           # entries (of ~1e4-6e4 candidates per tensor) allowed between REL_TOL and REL_TOL_DOWNSTREAM (measured: 0, once 1)
_report = {}


def _look(kind):
    sc = scenes.make_scene("human", P, seed=42)
    if kind == "trained":
        scenes.trained_look(sc, seed=7)
    return sc


def _upstream(seed, V=1):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(V, 3, H, W)).astype(np.float32), rng.normal(size=(V, 1, H, W)).astype(np.float32),
            rng.normal(size=(V, 1, H, W)).astype(np.float32))


def _compare(tag, name, ours, ref, floor=0.0, ref_end_to_end=None, skip_rows=None, loose_rows=None, max_loose=MAX_LOOSE_ENTRIES):
    """`ref`: oracle backward on the alpha image of the HIP forward (element-wise bar + max-normalised bar);
    `ref_end_to_end`: oracle backward on the oracle's own forward (max-normalised bar only);
    `skip_rows` [P] bool: knife-edge Gaussians, excluded from the element-wise bar."""
    ours = ours.detach().cpu().numpy().reshape(ref.shape).astype(np.float64)
    ref = ref.astype(np.float64)
    top = max(float(np.abs(ref).max()), floor) + 1e-30
    err = np.abs(ours - ref)
    e_max = float(err.max() / top)
    big = np.abs(ref) > FLOOR_FRAC * top
    if skip_rows is not None:
        big[skip_rows] = False
    if loose_rows is not None:          # rows behind a knife-edge subject: looser element-wise bar
        lb = big.copy()
        lb[~loose_rows] = False
        rl = err[lb] / np.abs(ref[lb]) if lb.any() else np.zeros(0)
        e_loose = float(rl.max()) if rl.size else 0.0
        assert e_loose < REL_TOL_DOWNSTREAM, "%s %s: per-element relative error %.3e behind a knife-edge subject" % (tag, name, e_loose)
        # the class is large (a few per cent of the Gaussians sit behind SOME knife-edge subject) but the looser bar is
        # needed by a handful of its entries only: at most MAX_LOOSE_ENTRIES may exceed the strict REL_TOL
        n_over = int((rl >= REL_TOL).sum())
        _report.setdefault(tag, {}).setdefault("_loose", {})[name] = dict(entries=int(lb.sum()), over_strict_bar=n_over, worst=e_loose)
        assert n_over <= max_loose, "%s %s: %d entries behind knife-edge subjects exceed REL_TOL" % (tag, name, n_over)
        big[loose_rows] = False
    e_rel = float((err[big] / np.abs(ref[big])).max()) if big.any() else 0.0
    rec = dict(max_norm=e_max, rel=e_rel, entries_checked=int(big.sum()), top=top)
    if big.any():                       # where the worst checked entry sits: which Gaussian, how large against the tensor maximum
        relmap = np.where(big, err / np.maximum(np.abs(ref), 1e-300), 0.0)
        idx = np.unravel_index(int(relmap.argmax()), relmap.shape)
        rec.update(worst_row=int(idx[0]), worst_ref=float(ref[idx]), worst_ours=float(ours[idx]), worst_ref_over_top=float(abs(ref[idx]) / top))
    if loose_rows is not None:
        rec["loose_rows"] = int(np.asarray(loose_rows).sum())
    line = "%-22s %-11s max-normalised %.2e   per-element relative %.2e on %d entries" % (tag, name, e_max, e_rel, big.sum())
    if ref_end_to_end is not None:
        e2e = float(np.abs(ours - ref_end_to_end.astype(np.float64)).max() / top)
        rec["max_norm_end_to_end"] = e2e
        line += "   end-to-end max-normalised %.2e" % e2e
        assert e2e < MAX_TOL, "%s %s: end-to-end max error / max |grad| = %.3e" % (tag, name, e2e)
    _report.setdefault(tag, {})[name] = rec
    print(line)
    assert e_max < MAX_TOL, "%s %s: max error / max |grad| = %.3e" % (tag, name, e_max)
    assert e_rel < REL_TOL, "%s %s: per-element relative error %.3e (entries above %.0e of the maximum)" % (
        tag, name, e_rel, FLOOR_FRAC)


def _dump():
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_headline.json"), "w") as f:
            json.dump(_report, f, indent=1)


@pytest.fixture
def list_mode(request, monkeypatch):
    """"culled" = the default lists (the fork's minus provably dead entries); "exact" = GipRasterConfig::exact_lists through
    GIP_RASTER_EXACT_LISTS=1: tiles_touched / num_rendered / key-value lists / ranges / n_contrib are the oracle's bit for
    bit (north_star: "tile/index buffers bit-exact") — driver-tested here at the headline size, not only at 10k / 256^2."""
    mode = getattr(request, "param", "culled")
    monkeypatch.setenv("GIP_RASTER_EXACT_LISTS", "1" if mode == "exact" else "0")
    return mode


@pytest.mark.parametrize("look,list_mode", [("init", "culled"), ("trained", "culled"), ("init", "exact"), ("trained", "exact")],
                         indirect=["list_mode"])
def test_single_view_forward_and_all_gradients_at_100k_1024(oracle, look, list_mode):
    from gaussianip_amd import GaussianRasterizer
    oracle.set_threads(oracle.max_threads())
    try:
        sc = _look(look)
        cam = scenes.train_cameras(4, 42, H, W)[0]
        bg = (0.0, 0.0, 0.0) if look == "init" else (0.2, 0.4, 0.1)
        Rn, ro = _check_forward_scene(oracle, sc, cam, H, W, 0, bg)        # integer buffers + images
        gC, gD, gA = _upstream(3)
        st = _settings(cam, H, W, bg, 0)
        t = {k: _dev(v).requires_grad_(True) for k, v in sc.items()}
        m2 = torch.zeros(P, 3, device="cuda", requires_grad=True)
        color, radii, depth, alpha = GaussianRasterizer(st)(means3D=t["means3D"], means2D=m2, opacities=t["opacities"],
                                                            shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        ((color * _dev(gC[0])).sum() + (depth * _dev(gD[0])).sum() + (alpha * _dev(gA[0])).sum()).backward()
        torch.cuda.synchronize()
        go_e2e = ro.backward(gC[0], gD[0], gA[0])
        go = ro.backward(gC[0], gD[0], gA[0], alpha_out=alpha.detach().cpu().numpy())
        knife, n_knife_pixels, behind = ro.knife_edge_gaussians(sharing=True)
    finally:
        oracle.set_threads(1)
    tag = "1 view / " + look + ("" if list_mode == "culled" else " / exact lists")
    print("%s: %d knife-edge pixels, %d Gaussians are the subject of a knife-edge test" % (tag, n_knife_pixels, int(knife.sum())))
    assert n_knife_pixels <= 5e-4 * H * W and knife.sum() <= 5e-3 * P, (n_knife_pixels, int(knife.sum()))
    print("%s: %d rows behind a knife-edge subject (REL_TOL_DOWNSTREAM class)" % (tag, int(behind.sum())))
    assert behind.sum() <= 0.10 * P, int(behind.sum())          # measured 6.0 % (init) / 8.6 % (trained); the guard is MAX_LOOSE_ENTRIES in _compare
    rot_floor = float(np.abs(go["scales"] * sc["scales"]).max())      # rotation of an isotropic splat: analytically 0
    for name, ours in (("means3D", t["means3D"].grad), ("means2D", m2.grad), ("opacities", t["opacities"].grad),
                       ("shs", t["shs"].grad), ("scales", t["scales"].grad), ("rotations", t["rotations"].grad)):
        _compare(tag, name, ours, go[name], floor=rot_floor if name == "rotations" else 0.0, ref_end_to_end=go_e2e[name],
                 skip_rows=knife, loose_rows=behind)
    _report[tag]["num_rendered"] = Rn
    _report[tag]["knife_edge_pixels"] = n_knife_pixels
    _report[tag]["knife_edge_gaussians"] = int(knife.sum())
    _dump()


@pytest.mark.parametrize("list_mode", ["culled", "exact"], indirect=True)
def test_four_view_launch_set_at_100k_1024(oracle, list_mode):
    """The training call: rasterize_views with the 4 cameras of one step.  Images per view against the oracle; parameter
    gradients against the float64 sum of the four oracle backwards; means2D gradients per view."""
    from gaussianip_amd import rasterize_views
    tag4 = "4 views" if list_mode == "culled" else "4 views / exact lists"
    sc = _look("init")
    cams = scenes.train_cameras(4, 42, H, W)
    bg = (0.0, 0.0, 0.0)
    gC, gD, gA = _upstream(5, V=4)
    sts = [_settings(c, H, W, bg, 0) for c in cams]
    t = {k: _dev(v).requires_grad_(True) for k, v in sc.items()}
    m2 = torch.zeros(4, P, 3, device="cuda", requires_grad=True)
    color, radii, depth, alpha = rasterize_views(t["means3D"], m2, t["opacities"], sts, shs=t["shs"], scales=t["scales"],
                                                 rotations=t["rotations"])
    ((color * _dev(gC)).sum() + (depth * _dev(gD)).sum() + (alpha * _dev(gA)).sum()).backward()
    torch.cuda.synchronize()
    alpha_np = alpha.detach().cpu().numpy()
    oracle.set_threads(oracle.max_threads())
    try:
        imgs, grads, ros = [], [], []
        for v, cam in enumerate(cams):
            ro, out = _oracle_forward(oracle, sc, cam, H, W, bg, 0)
            imgs.append(out)
            ros.append(ro)
            grads.append(ro.backward(gC[v], gD[v], gA[v], alpha_out=alpha_np[v]))
        kd = [ro.knife_edge_gaussians(sharing=True) for ro in ros]
        knife, behind = [k[0] for k in kd], [k[2] for k in kd]     # per view: subjects of a knife-edge test / rows behind one
    finally:
        oracle.set_threads(1)
        knife_wide = [ro.knife_edge_gaussians(thresh=1e-3)[0] for ro in ros]      # diagnostic: near-threshold at a 50x wider margin
    knife_any, behind_any = np.logical_or.reduce(knife), np.logical_or.reduce(behind)
    assert knife_any.sum() <= 1e-2 * P, int(knife_any.sum())       # ~1e-3 of the Gaussians per view
    # the looser REL_TOL_DOWNSTREAM class (rows blended behind a knife-edge subject) is bounded too
    print("4 views: rows behind a knife-edge subject per view %s, union %d" % ([int(b.sum()) for b in behind], int(behind_any.sum())))
    # measured: 6.0 / 12.4 / 13.2 / 14.3 % per view, 36.9 % in the union of the four views
    assert all(b.sum() <= 0.16 * P for b in behind) and behind_any.sum() <= 0.40 * P, [int(b.sum()) for b in behind]
    for v in range(4):
        o_color, o_radii, o_depth, o_alpha = imgs[v]
        assert np.array_equal(radii[v].cpu().numpy(), o_radii), "radii of view %d" % v
        _assert_images(ros[v], color[v], depth[v], alpha[v], o_color, o_depth, o_alpha)
        _compare(tag4, "means2D[%d]" % v, m2.grad[v], grads[v]["means2D"], skip_rows=knife[v], loose_rows=behind[v])
        rec = _report[tag4]["means2D[%d]" % v]
        if "worst_row" in rec:          # is the worst element-wise entry a NEAR knife-edge Gaussian (margin 2e-5 .. 1e-3)?
            rec["worst_row_near_knife_edge_1e-3"] = bool(knife_wide[v][rec["worst_row"]])
            rec["worst_row_shares_a_knife_edge_pixel"] = bool(behind[v][rec["worst_row"]])
            print("   worst row %d: ref %.3e (%.1e of the maximum), ours %.3e, near-knife-edge(1e-3) %s" % (
                rec["worst_row"], rec["worst_ref"], rec["worst_ref_over_top"], rec["worst_ours"], rec["worst_row_near_knife_edge_1e-3"]))
    tot = {k: sum(g[k].astype(np.float64) for g in grads) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    rot_floor = float(np.abs(tot["scales"] * sc["scales"]).max())
    for k in ("means3D", "opacities", "shs", "scales"):
        _compare(tag4, k, t[k].grad, tot[k], skip_rows=knife_any, loose_rows=behind_any)
    _compare(tag4, "rotations", t["rotations"].grad, tot["rotations"], floor=rot_floor, skip_rows=knife_any, loose_rows=behind_any)
    _dump()
