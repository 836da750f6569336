"""``fastvision`` -- alias of ``fastvision_amd`` under the reference's own package name.

The reference is imported as ``fastvision`` (its checkout directory is the package: generate/template-yolov3/train.py:68-70
``from fastvision.detection.models import yolov3`` ...).  With this directory on the path those lines resolve, UNCHANGED, to the
MI355X implementation: every ``fastvision.x.y`` is the very module object ``fastvision_amd.x.y`` (one copy of every module,
hence one copy of the library state), not a second import of the same file under another name.

Only the hot path exists on this side: ``fastvision.classfication.models.darknet53``,
``fastvision.detection.{neck,head,models,tools}``, ``fastvision.loss``, ``fastvision.metrics``, ``fastvision.datasets``,
``fastvision.utils``; importing anything else of the reference's tree raises ModuleNotFoundError.
"""
import importlib
import pkgutil
import sys

import fastvision_amd as _impl

_SKIP = ('fastvision_amd.demos', 'fastvision_amd.csrc')


def _alias_all():
    this = sys.modules[__name__]
    names = ['fastvision_amd'] + [m.name for m in pkgutil.walk_packages(_impl.__path__, 'fastvision_amd.')
                                  if not m.name.startswith(_SKIP)]
    for real in names:
        mod = importlib.import_module(real)
        alias = 'fastvision' + real[len('fastvision_amd'):]
        if alias != 'fastvision':
            sys.modules[alias] = mod
    for k, v in vars(_impl).items():               # top-level names: sub-packages, compute_dtype, FusedAdam, ...
        if not k.startswith('__'):
            setattr(this, k, v)


_alias_all()
__version__ = _impl.__version__
