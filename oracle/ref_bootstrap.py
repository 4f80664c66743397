"""Container-only bootstrap that makes the upstream GIGA reference importable.

TEST INFRASTRUCTURE -- never imported by the product (`giga_amd/`), never run on the GPU box
(`/root/reference` does not exist there).  Used only by `oracle/make_goldens.py` and by the
`reference`-marked CPU tests, which skip themselves when `/root/reference` is absent.

The reference (`/root/reference/src/vgn`) does not import as shipped in this image because of
missing third-party packages that are irrelevant to the hot path:
  * `torch_scatter`  (encoder/voxels.py:4, encoder/pointnet.py:5)  -> stub with `scatter_mean`
    implemented as scatter_add / clamp(count, 1), the published torch-scatter==2.0.6 semantics
    (environment.yaml:145).  `scatter_max` is never reached on the GIGA path.
  * `torchvision`, `trimesh`, `PIL`, `skimage` ... -> empty stubs (mesh / dataset code only).
  * `np.int` & friends removed in numpy 2 (utils/binvox_rw.py:206) -> aliases.
Nothing here restates reference code: it only supplies absent *dependencies*.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_SRC, "vgn"))


def _scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    import torch

    if dim < 0:
        dim = src.dim() + dim
    index = index.expand_as(src) if index.shape != src.shape else index
    if out is None:
        size = list(src.shape)
        size[dim] = int(index.max()) + 1 if dim_size is None else dim_size
        out = src.new_zeros(size)
    out.scatter_add_(dim, index, src)
    count = torch.zeros_like(out)
    count.scatter_add_(dim, index, torch.ones_like(src))
    out.div_(count.clamp_(min=1))
    return out


def _scatter_max(*a, **k):  # pragma: no cover - not on the GIGA path
    raise NotImplementedError("scatter_max is not on the GIGA hot path")


class _Anything(types.ModuleType):
    """A module whose every attribute is another permissive stub (for `from x import y`)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Anything(self.__name__ + "." + name)
        sub.__path__ = []
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return None


_STUB_ROOTS = ("torchvision", "trimesh", "PIL", "skimage", "open3d", "pybullet", "urdfpy",
               "ignite", "tensorboard", "catkin_pkg", "pykdtree", "matplotlib", "plyfile",
               "mcubes", "pyrender", "mpl_toolkits")


class _StubFinder:
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Anything(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install():
    """Idempotently install stubs + numpy aliases and put the reference on sys.path."""
    if not reference_available():
        raise RuntimeError("reference not present at %s (expected on the build container only)"
                           % REFERENCE_SRC)
    import numpy as np

    for name, typ in (("int", int), ("float", float), ("bool", bool), ("object", object),
                      ("long", int), ("complex", complex)):
        if name not in np.__dict__:
            setattr(np, name, typ)
    if "torch_scatter" not in sys.modules:
        ts = types.ModuleType("torch_scatter")
        ts.scatter_mean = _scatter_mean
        ts.scatter_max = _scatter_max
        sys.modules["torch_scatter"] = ts
    real = set()
    for root in _STUB_ROOTS:
        try:
            if importlib.util.find_spec(root) is not None:
                real.add(root)
        except (ImportError, ValueError):
            pass
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        finder = _StubFinder()
        sys.meta_path.append(finder)  # appended: real packages (if any) win
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)


def load_reference_giga(state_dict=None, name="giga"):
    """Build the reference network (`vgn.networks.get_network`) and optionally load weights."""
    install()
    import importlib.util  # noqa: F401
    from vgn.networks import get_network  # type: ignore

    net = get_network(name)
    if state_dict is not None:
        net.load_state_dict(state_dict)
    return net.eval()
