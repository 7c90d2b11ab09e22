"""Authoring-time harness: imports the REAL reference (read-only checkout at /root/reference) on CPU so that
oracle/gen_golden.py can record golden vectors and validate the restatement.  TEST INFRASTRUCTURE.

It never copies reference source: it only registers empty stand-ins for third-party packages this image
lacks (munch, skimage, trimesh, pytorch3d, ...) -- none of which are on the rendering path -- and stops
torch.utils.cpp_extension.load from JIT-compiling the reference's CUDA files, so that the reference's own
PyTorch CPU fallbacks of the two custom ops run.  Does nothing useful on the GPU box
(/root/reference does not exist there); nothing outside oracle/gen_golden.py and the tests marked
`needs_reference` calls it."""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("E3DGE_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "project"))


class _Anything:
    """Attribute sink used for names the reference imports but the rendering path never touches."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__getattr__ = lambda attr: _Anything        # PEP 562: any other name resolves to a sink
    m.__path__ = []                               # behave like a package for `import a.b`
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent:
        setattr(_stub(parent), child, m)
    return m


class Munch(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self):
        return Munch(dict.copy(self))


_prepared = False


def prepare():
    """chdir into the reference (it appends the RELATIVE path project/vendor/pifu, volume_renderer.py:15),
    register the stand-ins, neutralise the CUDA JIT."""
    global _prepared
    if _prepared:
        return
    if not available():
        raise RuntimeError(f"reference checkout not found at {REF_ROOT}")
    import numpy as np
    import torch
    import torch.utils.cpp_extension as cpp_ext
    sys.dont_write_bytecode = True
    os.chdir(REF_ROOT)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    names = ['munch', 'skimage', 'skimage.measure', 'trimesh', 'pytorch3d', 'pytorch3d.renderer', 'pytorch3d.structures',
             'pytorch3d.transforms', 'pytorch3d.renderer.mesh', 'pytorch3d.io', 'pytorch3d.ops', 'omegaconf',
             'omegaconf.dictconfig', 'cv2', 'IPython', 'IPython.display', 'ipdb', 'torchvision',
             'torchvision.transforms', 'torchvision.models', 'torchvision.models.resnet', 'torchvision.models.vgg',
             'torchvision.utils', 'torchvision.transforms.functional', 'kornia', 'lmdb', 'wandb', 'mmcv',
             'mmcv.utils', 'facexlib', 'skvideo', 'skvideo.io', 'configargparse', 'sorcery', 'imageio', 'lpips']
    tops = sorted({n.split('.')[0] for n in names})
    missing = {t for t in tops if t not in sys.modules and importlib.util.find_spec(t) is None}
    for n in names:
        if n.split('.')[0] in missing:
            _stub(n)
    if 'munch' in missing:
        sys.modules['munch'].Munch = Munch
    if not hasattr(np, 'deprecate'):
        np.deprecate = lambda *a, **k: (lambda f: f)                # vendor/pifu/lib/geometry.py:1,7
    cpp_ext.load = lambda *a, **k: _Anything()                      # no CUDA JIT; CPU tensors use the fallbacks
    _prepared = True


def modules():
    """(volume_renderer, stylesdf_model, camera_utils, op) of the reference."""
    prepare()
    vr = importlib.import_module('project.utils.volume_renderer')
    sm = importlib.import_module('project.models.stylesdf_model')
    cu = importlib.import_module('project.utils.camera_utils')
    op = importlib.import_module('project.models.op')
    return vr, sm, cu, op
