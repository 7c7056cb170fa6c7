"""Golden vectors for the FFL loss (models/ffl/losses.py build_combined_loss + MultiLoss) produced by the REFERENCE's own module.

Runs only in the build container.  Third-party packages the reference imports but that are absent here (kornia, skimage, shapely,
cv2, rasterio, ...) are replaced by empty auto-stubs so that the modules written in the reference repo import; the one function the
loss path really takes from kornia, `normalize_kernel2d` (kernel / sum |kernel|, two lines), is supplied - parity of that detail is
"unpinned vs kornia".  Usage: python tests/golden/make_ffl_loss_golden.py"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF + "/ffl_submodules/pytorch_lydorn")
sys.path.insert(0, REF + "/ffl_submodules/lydorn_utils")


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (object,), {})


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("kornia", "skimage", "shapely", "cv2", "rasterio", "laspy", "pycocotools", "matplotlib", "descartes", "fiona", "pyproj",
             "overpy", "numba", "jsmin")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def load_losses():
    sys.meta_path.append(_Finder())
    base = REF + "/pixelspointspolygons"
    for name, path in (("pixelspointspolygons", base), ("pixelspointspolygons.models", base + "/models"),
                       ("pixelspointspolygons.models.ffl", base + "/models/ffl")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    import kornia.filters.kernels as kk

    def normalize_kernel2d(inp):
        norm = inp.abs().sum(dim=-1).sum(dim=-1)
        return inp / norm[..., None, None]
    kk.normalize_kernel2d = normalize_kernel2d
    import pixelspointspolygons.models.ffl.losses as L
    return L


def ns(**kw):
    return types.SimpleNamespace(**kw)


def make_cfg(use_freq=False, use_dist=False, use_size=False):
    """the fields build_combined_loss reads, with config/model/ffl.yaml's values"""
    loss = ns(multi=ns(epoch_thresholds=[0, 5, 10],
                       weights=ns(seg=1, crossfield_align=1, crossfield_align90=0.5, crossfield_smooth=0.005,
                                  seg_interior_crossfield=[0, 0, 0.2], seg_edge_crossfield=[0, 0, 0.2], seg_edge_interior=[0, 0, 0.2])),
              seg=ns(bce_coef=1.0, dice_coef=0.2, use_freq=use_freq, use_dist=use_dist, use_size=use_size, w0=50, sigma=10, type="bool"))
    model = ns(compute_seg=True, compute_crossfield=True, seg=ns(compute_interior=True, compute_edge=False, compute_vertex=False), loss=loss)
    return ns(experiment=ns(model=model, encoder=ns(in_height=224, in_width=224)), host=ns(device="cpu"))


def make_case(B, H, seed):
    g = torch.Generator().manual_seed(seed)
    seg = torch.sigmoid(torch.randn(B, 1, H, H, generator=g) * 2.0)
    cf = 2.0 * torch.tanh(torch.randn(B, 4, H, H, generator=g))
    gt = torch.rand(B, 3, H, H, generator=g)
    gt[:, 0] = (gt[:, 0] > 0.6).float() * (0.9 + 0.1 * torch.rand(B, H, H, generator=g))      # interior: values around the 0.98 threshold
    gt[:, 1] = (gt[:, 1] > 0.8).float() * torch.rand(B, H, H, generator=g)
    gt[:, 2] = (gt[:, 2] > 0.95).float() * torch.rand(B, H, H, generator=g)
    angle = (torch.rand(B, 1, H, H, generator=g) * 2 - 1) * np.pi
    return seg, cf, gt, angle


def main():
    L = load_losses()
    crit = L.build_combined_loss(make_cfg())
    names = [f.name for f in crit.loss_funcs]
    arrays = {"names": np.array(names)}
    cases = [("s16", 2, 16, 1, 0.0, None), ("s16e3", 2, 16, 2, 3.0, None), ("s33e12", 3, 33, 3, 12.0, [0.7, 0.05, 0.3, 2.0, 0.01]),
             ("s96e7", 1, 96, 4, 7.5, [1.3, 0.02, 0.04, 0.5, 0.2])]
    for tag, B, H, seed, epoch, norms in cases:
        seg, cf, gt, angle = make_case(B, H, seed)
        for f, nv in zip(crit.loss_funcs, norms or [1.0] * len(names)):
            f.norm[0] = nv
        seg.requires_grad_(True)
        cf.requires_grad_(True)
        gt_batch = {"gt_polygons_image": gt, "gt_crossfield_angle": angle, "distances": torch.ones(B, 1, H, H), "sizes": torch.ones(B, 1, H, H),
                    "class_freq": torch.full((B, 3), 0.2)}
        total, ind, _ = crit({"seg": seg, "crossfield": cf}, gt_batch, normalize=True, epoch=epoch)
        total.backward()
        arrays.update({f"{tag}::seg": seg.detach().numpy(), f"{tag}::crossfield": cf.detach().numpy(), f"{tag}::gt": gt.numpy(),
                       f"{tag}::angle": angle.numpy(), f"{tag}::epoch": np.float64(epoch), f"{tag}::norms": np.array(norms or [1.0] * len(names)),
                       f"{tag}::total": total.detach().numpy(), f"{tag}::losses": np.array([float(ind[n]) for n in names]),
                       f"{tag}::dseg": seg.grad.numpy(), f"{tag}::dcf": cf.grad.numpy()})
    # pixel-weighted BCE (use_freq + use_dist + use_size), reference pre-process compute_seg_loss_weigths
    critw = L.build_combined_loss(make_cfg(True, True, True))
    seg, cf, gt, angle = make_case(2, 24, 9)
    g = torch.Generator().manual_seed(99)
    dist = torch.rand(2, 1, 24, 24, generator=g) * 0.05
    sizes = 0.01 + torch.rand(2, 1, 24, 24, generator=g) * 0.2
    cfreq = torch.tensor([[0.3, 0.1, 0.02], [0.2, 0.05, 0.01]])
    seg.requires_grad_(True)
    cf.requires_grad_(True)
    total, ind, _ = critw({"seg": seg, "crossfield": cf}, {"gt_polygons_image": gt, "gt_crossfield_angle": angle, "distances": dist, "sizes": sizes,
                                                          "class_freq": cfreq}, normalize=True, epoch=6.0)
    total.backward()
    arrays.update({"w24::seg": seg.detach().numpy(), "w24::crossfield": cf.detach().numpy(), "w24::gt": gt.numpy(), "w24::angle": angle.numpy(),
                   "w24::distances": dist.numpy(), "w24::sizes": sizes.numpy(), "w24::class_freq": cfreq.numpy(), "w24::epoch": np.float64(6.0),
                   "w24::total": total.detach().numpy(), "w24::losses": np.array([float(ind[n]) for n in names]),
                   "w24::dseg": seg.grad.numpy(), "w24::dcf": cf.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "ffl_loss.npz"), **arrays)
    print("wrote ffl_loss.npz", names, os.path.getsize(os.path.join(HERE, "ffl_loss.npz")))


if __name__ == "__main__":
    main()
