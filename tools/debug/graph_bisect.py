"""GPU box: which piece of the hot path breaks HIP-graph capture?  Each case is captured + replayed in its own process
(a failing hipStreamEndCapture takes the process down).  Usage: graph_bisect.py            -> runs every case
                                                                  graph_bisect.py <case>     -> runs one, prints RESULT"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CASES = ["fullterms_16_256"]

def main_one(case):
    import torch
    from tests.helpers import scene
    from umr_amd import functional as UF, loss_utils as LU, geom_utils as GU
    from umr_amd.smr import SoftRenderer
    DEV = "cuda:0"
    torch.manual_seed(0)
    verts, faces, cams, gen = scene(2, 2, seed=1)
    verts = verts.to(DEV).requires_grad_(True); cams = cams.to(DEV).requires_grad_(True); faces = faces.to(DEV)
    F = faces.shape[1]
    tex = torch.rand(2, F, 36, 3, device=DEV, requires_grad=True)
    imgs = torch.rand(2, 3, 64, 64, device=DEV); masks = (torch.rand(2, 64, 64, device=DEV) > 0.5).float()
    flow = (torch.rand(2, F, 6, 6, 2, device=DEV) * 2 - 1).requires_grad_(True)
    leaves = [verts, cams, tex, flow]
    r_sil = SoftRenderer(64); r_tex = SoftRenderer(64); r_tex.ambient_light_only()
    hard_r = SoftRenderer(64, "hard"); hard_r.ids_only = True
    angles = torch.tensor([10., 20.], device=DEV); dts = torch.rand(2, 1, 64, 64, device=DEV)
    class _H(torch.nn.Module):
        pass
    holder = _H()
    if case.startswith("stage"):
        from umr_amd.synthetic import template
        tv, fc = template(2); lap = LU.LaplacianLoss(tv, fc).to(DEV); flat = LU.FlattenLoss(fc).to(DEV)
    if case.startswith("full"):
        from umr_amd.synthetic import make_s1_inputs
        from umr_amd.train_step import RenderCompareS1
        from umr_amd.perceptual import PerceptualTextureLoss
        if case.startswith("fullterms"):
            bb, hh = int(case.split("_")[1]), int(case.split("_")[2])
        else:
            bb, hh = (4 if case == "full_b4" else 2), 64
        tv, fc, outputs, batch = make_s1_inputs(bb, hh, 3 if hh == 256 else 2, seed=3, device=DEV)
        rc = RenderCompareS1(tv.to(DEV), fc.to(DEV), hh, texture_loss=PerceptualTextureLoss(DEV) if (case == "full" or (case.startswith("fullterms") and "noperc" not in case)) else None).to(DEV)
        TERMS = {}
        leaves = [outputs["delta_v"], outputs["cam"], outputs["tex_flow"]]
        if case == "full_leafverts":
            outputs["pred_vs"] = (outputs["mean_shape"][None] + outputs["delta_v"]).detach().requires_grad_(True)
            leaves = [outputs["pred_vs"], outputs["cam"], outputs["tex_flow"]]
        if case == "full_sumterms":
            import umr_amd.train_step as TS
            TS.weighted_total = lambda module, terms, weights: sum(terms[k] * w for k, w in weights)
    if case == "alexnet":
        from umr_amd.perceptual import PerceptualTextureLoss
        ptl = PerceptualTextureLoss(DEV)
        pimg = torch.rand(2, 3, 64, 64, device=DEV, requires_grad=True); leaves = [pimg]
    if case == "laplacian":
        from umr_amd.synthetic import template
        tv, fc = template(2); lap = LU.LaplacianLoss(tv, fc).to(DEV); flat = LU.FlattenLoss(fc).to(DEV)
    if case == "flatten":
        from umr_amd.synthetic import template
        tv, fc = template(2); flat = LU.FlattenLoss(fc).to(DEV)

    def step():
        for l in leaves:
            l.grad = None
        if case == "torch_only":
            y = (verts * 2).sin().sum() + cams.pow(2).sum(); y.backward(); return y
        if case == "project":
            _, fo, _ = UF.ProjectFacesFunction.apply(verts, cams, faces.int(), 5.0, -2.732, False); y = fo.sum(); y.backward(); return y
        if case == "sil_fwd":
            with torch.no_grad():
                return r_sil.silhouettes(verts, faces, cams).sum()
        if case == "sil_fwd_bwd":
            y = r_sil.silhouettes(verts, faces, cams).sum(); y.backward(); return y
        if case == "raster_fwd":
            with torch.no_grad():
                return r_tex(verts, faces, cams, tex)[0].sum()
        if case == "raster_fwd_bwd":
            y = r_tex(verts, faces, cams, tex)[0].sum(); y.backward(); return y
        if case == "grid_sample":
            y = GU.sample_textures(flow, imgs).sum(); y.backward(); return y
        if case == "neg_iou":
            a = r_sil.silhouettes(verts, faces, cams); y = LU.neg_iou_loss(a, masks); y.backward(); return y
        if case == "laplacian":
            y = lap(verts).mean(); y.backward(); return y
        if case == "flatten":
            y = flat(verts).mean(); y.backward(); return y
        if case == "texcycle":
            with torch.no_grad():
                hr = SoftRenderer(64, "hard"); hr.ids_only = True
                _, p2f, aggr = hr(verts, faces, cams)
            y, _ = LU.TexCycle()(flow, p2f, aggr[:, 1].reshape(2, -1)); y.backward(); return y
        if case == "cos_sim":
            from umr_amd.perceptual import cos_sim_distance
            y = cos_sim_distance([tex.view(2, F, 108).permute(0, 2, 1).reshape(2, 108, 16, F // 16).contiguous()],
                                 [(tex * 0.5 + 0.1).view(2, F, 108).permute(0, 2, 1).reshape(2, 108, 16, F // 16).contiguous()]).sum()
            y.backward(); return y
        if case == "alexnet":
            y = ptl(pimg, imgs, masks, masks); y.backward(); return y
        if case == "rotate":
            from umr_amd.train_step import rotate_cam_y
            return rotate_cam_y(cams.detach(), angles).sum()
        if case == "wtotal":
            from umr_amd.train_step import weighted_total
            t = {"a": verts.pow(2).mean(), "b": cams.abs().mean(), "c": tex.mean()}
            y = weighted_total(holder, t, [("a", 1.0), ("b", 0.5), ("c", 0.0)]); y.backward(); return y
        if case.startswith("stage"):
            from umr_amd.train_step import rotate_cam_y, weighted_total
            lvl = "ABCDE".index(case[-1])
            B = 2
            fcs = faces
            terms = {}
            random_cams = rotate_cam_y(cams.detach(), angles)
            both = r_sil.silhouettes(verts, fcs, torch.stack((cams, random_cams), dim=1).reshape(2 * B, 7)).view(B, 2, 64, 64)
            seen, unseen = both[:, 0], both[:, 1]
            terms["mask"] = LU.neg_iou_loss(seen, masks)
            if lvl >= 1:
                terms["triangle"] = lap(verts).mean(); terms["flatten"] = flat(verts).mean()
                terms["deform"] = LU.deform_l2reg(verts); terms["ori"] = LU.sym_reg(verts)
            if lvl >= 2:
                t = GU.sample_textures(flow, imgs).reshape(B, F, -1, 3)
                rgba, p2f, _ = r_tex(verts.detach(), fcs, cams.detach(), t)
                terms["tex"] = LU.texture_loss_masks(rgba[:, 0:3], imgs, masks, seen)
            if lvl >= 3:
                terms["tex_dt"] = LU.texture_dt_loss(flow, dts)
                _, _, aggr = hard_r(verts.detach(), fcs, cams.detach())
                terms["tex_cycle"], _ = LU.TexCycle()(flow, p2f.detach(), aggr[:, 1].reshape(B, -1).detach())
            if lvl >= 4:
                terms["gan"] = unseen.mean()
                y = weighted_total(holder, terms, [(k, 1.0) for k in terms])
            else:
                y = sum(terms.values())
            y.backward(); return y
        if case == "full_members":
            from umr_amd.train_step import rotate_cam_y
            B = 2
            pv = outputs["mean_shape"][None] + outputs["delta_v"]
            fcs = rc.faces[None].expand(B, -1, -1)
            cam = outputs["cam"]
            random_cams = rotate_cam_y(cam.detach(), batch["gan_angles"])
            both = rc.renderer.silhouettes(pv, fcs, torch.stack((cam, random_cams), dim=1).reshape(2 * B, 7)).view(B, 2, 64, 64)
            y = LU.neg_iou_loss(both[:, 0], batch["masks"]) + rc.laplacian_loss_fn(pv).mean() + rc.flatten_loss_fn(pv).mean()
            t = GU.sample_textures(outputs["tex_flow"], batch["imgs"]).reshape(B, fcs.shape[1], -1, 3)
            rgba, p2f, _ = rc.tex_renderer(pv.detach(), fcs, cam.detach(), t)
            y = y + LU.texture_loss_masks(rgba[:, 0:3], batch["imgs"], batch["masks"], both[:, 0]) + both[:, 1].mean()
            y.backward(); return y
        if case.startswith("full"):
            if case != "full_leafverts":
                outputs["pred_vs"] = outputs["mean_shape"][None] + outputs["delta_v"]
            if case == "full_fwdonly":
                with torch.no_grad():
                    total, _ = rc(outputs, batch)
                return total
            total, tt = rc(outputs, batch); total.backward()
            if case.startswith("fullterms"):
                TERMS.update(tt)
            return total
        raise SystemExit("unknown case")

    eager = float(step())
    eager_terms = {k: float(v) for k, v in TERMS.items()} if case.startswith("fullterms") else {}
    eager_grads = [l.grad.detach().clone() for l in leaves] if case.startswith("fullterms") else []
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    if case.startswith("full") and case != "full_leafverts":
        outputs["pred_vs"] = None
        TERMS.clear()
    for l in leaves:
        l.grad = None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capturing", case, flush=True)
    with torch.cuda.graph(g):
        out = step()
    print("captured", case, flush=True)
    g.replay(); torch.cuda.synchronize()
    if case.startswith("fullterms"):
        print("terms eager vs replay:", {k: (round(eager_terms[k], 6), round(float(TERMS[k]), 6)) for k in eager_terms}, flush=True)
        print("grad max rel diff:", [float((a - l.grad).abs().max() / a.abs().max()) for a, l in zip(eager_grads, leaves)], flush=True)
        for r in range(2, 61):
            g.replay()
            if r in (2, 5, 10, 20, 40, 60):
                torch.cuda.synchronize()
                print("replay %d terms:" % r, {k: round(float(TERMS[k]), 5) for k in eager_terms}, "total", float(out), flush=True)
    print("RESULT %s ok eager %.6f replay %.6f" % (case, eager, float(out)), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        main_one(sys.argv[1])
    else:
        for c in CASES:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=300)
            res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
            print("\n".join(l for l in p.stdout.splitlines() if l.startswith(("terms", "grad", "replay"))), flush=True)
            err = [l for l in p.stderr.splitlines() if ("Error" in l or "error" in l) and "amdgpu.ids" not in l][-2:]
            print(res[0] if res else "RESULT %s FAILED rc=%d last=%s err=%s" % (c, p.returncode, p.stdout.strip().splitlines()[-1:] , err), flush=True)
