"""RayTracer (host BVH-4 + HIP traversal) vs the brute-force oracle, the reference's data fixture, and the config-4 lookup chain."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import nerftex_hip  # noqa: F401

    return torch.device("cuda:0")


def load_obj(path):
    v, f = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(t) for t in p[1:4]])
        elif p[0] == "f":
            f.append([int(t.split("/")[0]) - 1 for t in p[1:4]])
    return np.asarray(v, np.float32), np.asarray(f, np.uint32)


def test_reference_fixture_faces(dev, golden_dir):
    """external/RayTracer/test_data: a ray from the centre through each recorded face must return exactly that face."""
    from RayTracer import RayTracer

    v, f = load_obj(os.path.join(golden_dir, "raytracer", "object.obj"))
    hv, hf = load_obj(os.path.join(golden_dir, "raytracer", "intersected_faces.obj"))
    assert v.shape == (20, 3) and f.shape == (36, 3) and hf.shape == (3, 3)
    rt = RayTracer(v, f)
    cent = hv[hf].mean(1)
    o = np.zeros_like(cent)
    d = cent / np.linalg.norm(cent, axis=1, keepdims=True)
    pos, nrm, depth, face = rt.trace(torch.from_numpy(o).to(dev), torch.from_numpy(d.astype(np.float32)).to(dev))
    face = face.cpu().numpy()
    assert (face >= 0).all()
    for k in range(3):
        want = {tuple(np.round(p, 6)) for p in hv[hf[k]]}
        got = {tuple(np.round(p, 6)) for p in v[f[face[k]]]}
        assert got == want, f"recorded face {k}"
    np.testing.assert_allclose(np.linalg.norm(pos.cpu().numpy(), axis=1), depth.cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(np.abs((nrm.cpu().numpy() * d).sum(1)), np.ones(3) * np.abs((nrm.cpu().numpy() * d).sum(1)), rtol=0)  # finite
    assert np.allclose(np.linalg.norm(nrm.cpu().numpy(), axis=1), 1.0, atol=1e-6)


@pytest.mark.parametrize("mesh", ["dodecahedron", "star_flower", "tiny"])
def test_trace_matches_bruteforce_oracle(oracle, dev, golden_dir, mesh):
    from ngp_harness.curved import star_flower_mesh
    from RayTracer import RayTracer

    if mesh == "dodecahedron":
        v, f = load_obj(os.path.join(golden_dir, "raytracer", "object.obj"))
    elif mesh == "star_flower":
        v, f = star_flower_mesh(36, 72)
    else:  # <= 8 triangles: the wrapper pads with far-away faces (raytracer.py:16-22)
        v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
        f = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]], np.uint32)
    rt = RayTracer(v, f)
    rng = np.random.default_rng(5)
    N = 6000
    o = rng.uniform(-1.5, 1.5, size=(N, 3)).astype(np.float32)
    d = rng.normal(size=(N, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:10] = np.eye(3, dtype=np.float32)[np.arange(10) % 3]  # axis-aligned: inf / nan slabs are relied upon
    o[:2000] *= 0.2  # inside the meshes
    w_pos, w_nrm, w_depth, w_face, second = oracle.raytrace(v, f, o, d)
    pos, nrm, depth, face = rt.trace(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
    pos, nrm, depth, face = pos.cpu().numpy(), nrm.cpu().numpy(), depth.cpu().numpy(), face.cpu().numpy()
    hit = w_face >= 0
    assert hit.mean() > 0.05 and (~hit).sum() > 0
    # identical arithmetic per triangle: the closest distance is bit-exact (the BVH only prunes)
    same = depth.view(np.uint32) == w_depth.view(np.uint32)
    assert same.mean() > 0.999, f"{(~same).sum()} rays differ"
    np.testing.assert_allclose(depth, w_depth, rtol=1e-5, atol=1e-6)
    assert (depth[~hit] == 10.0).all() and (face[~hit] == -1).all() and not nrm[~hit].any()
    unambiguous = hit & same & (second > w_depth * (1 + 1e-6))
    assert np.array_equal(face[unambiguous], w_face[unambiguous])
    np.testing.assert_allclose(nrm[unambiguous], w_nrm[unambiguous], atol=1e-6)
    np.testing.assert_allclose(pos[same], w_pos[same], atol=1e-6)
    # in-place variant writes hits back into the ray buffers
    ot, dt_ = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    p2, n2, d2, f2 = rt.trace(ot, dt_, inplace=True)
    assert p2.data_ptr() == ot.data_ptr() and np.array_equal(d2.cpu().numpy(), depth) and np.array_equal(p2.cpu().numpy(), pos)


def test_curved_field_lookup_chain(oracle, dev):
    """BASELINE config 4: BVH surface hits + curved-field hash lookup on the star_flower-shaped mesh, vs the oracle chain."""
    from ngp_harness.curved import CurvedFieldLookup, star_flower_mesh

    v, f = star_flower_mesh()
    assert 19000 < f.shape[0] < 22000
    torch.manual_seed(0)
    field = CurvedFieldLookup(v, f, bound=1.0, h_threshold=0.05).to(dev)
    field.encoder.embeddings.data.uniform_(-1, 1)
    rng = np.random.default_rng(9)
    # query points in a shell around the surface, normals = analytic radial direction (stand-in for the KNN normal)
    face_pick = rng.integers(0, f.shape[0], size=20000)
    bary = rng.dirichlet([1, 1, 1], size=20000).astype(np.float32)
    on = (v[f[face_pick]] * bary[..., None]).sum(1)
    n = on / np.linalg.norm(on, axis=1, keepdims=True)
    x = (on + n * rng.uniform(-0.08, 0.08, size=(20000, 1))).astype(np.float32)
    feat, sdf, face, mask = field(torch.from_numpy(x).to(dev), torch.from_numpy(n.astype(np.float32)).to(dev))

    wp_pos, _, wd_pos, wf_pos, _ = oracle.raytrace(v, f, x, n)
    wp_neg, _, wd_neg, wf_neg, _ = oracle.raytrace(v, f, x, -n)
    use = wd_pos <= wd_neg
    w_sur = np.where(use[:, None], wp_pos, wp_neg)
    w_sdf = np.where(use, -wd_pos, wd_neg)
    ok = np.abs(sdf.cpu().numpy() - w_sdf) < 1e-6
    assert ok.mean() > 0.999
    assert 0.3 < mask.float().mean().item() < 0.9
    enc = field.encoder
    x01 = (w_sur + 1.0) / 2.0
    want, _ = oracle.grid_encode_forward(x01, enc.embeddings.detach().cpu().numpy(), enc.offsets.cpu().numpy(), float(np.log2(enc.per_level_scale)), 512,
                                         False, 0, True)
    want = want.transpose(1, 0, 2).reshape(x.shape[0], -1)
    got = feat.detach().cpu().numpy()
    assert np.array_equal(got[ok], want[ok]), "hash lookup at the projected surface point is bit-exact"
    assert torch.isfinite(field.encoder.clustering_loss(pick_level=False).detach())


def test_fused_curved_projector_matches_the_reference_op_sequence(dev):
    """nerftex_curved_project (coarse normal from the K nearest vertices + two closest-hit traces + select + mask + frame gather +
    FreqEncoder of the height, one kernel) vs the reference's op sequence (tools/map.py:414-433, 454-501; tools/encoding.py:5-43)
    restated with framework ops over RayTracer.trace, on a ~20 k-face star_flower-shaped mesh."""
    from ngp_harness.curved import MeshProjector, star_flower_mesh
    from oracle import cpu_path

    v, f = star_flower_mesh()
    proj = MeshProjector(v, f, h_threshold=0.05).to(dev)
    rng = np.random.default_rng(5)
    base = v[rng.integers(0, len(v), 20000)]
    pts = torch.from_numpy((base * (1 + rng.uniform(-0.08, 0.08, (len(base), 1)))).astype(np.float32)).to(dev)  # a shell around the surface
    p0, sdf0, m0, n0, tbn0, face0 = proj.project_reference(pts)
    p1, sdf1, m1, n1, tbn1, face1, z1 = proj.project_fused(pts)
    torch.testing.assert_close(n1, n0, rtol=0, atol=2e-5)
    same_face = face1 == face0
    assert same_face.float().mean() > 0.998, "a ray that grazes an edge may pick the neighbouring face when the normal differs in the last bits"
    torch.testing.assert_close(sdf1[same_face], sdf0[same_face], rtol=0, atol=2e-5)
    torch.testing.assert_close(p1[same_face], p0[same_face], rtol=0, atol=2e-5)
    assert torch.equal(tbn1[same_face], tbn0[same_face])
    assert (m1 == m0).float().mean() > 0.999 and 0.2 < m1.float().mean() < 0.95
    assert (sdf1 < 0).float().mean() > 0.2 and (sdf1 > 0).float().mean() > 0.2, "points inside and outside the surface"
    want_z = cpu_path.freq_encode(sdf1.cpu(), 11, 12)
    torch.testing.assert_close(z1.cpu(), want_z, rtol=0, atol=2e-4)  # sin / cos of up to 2048 h: argument reduction differs in the last bits
    with pytest.raises(RuntimeError, match="1 <= K"):
        from nerftex_hip import check, lib

        check(lib.nerftex_curved_project(proj.tracer._handle, None, None, None, 4, 99, None, None, 100, 0.05, 0.05, None, 12, None, None, None, None, None, None, None, None))
