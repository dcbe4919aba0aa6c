from . import functional  # noqa: F401


def _unavailable(name):
    class _T:
        def __init__(self, *a, **k):
            raise NotImplementedError(f'torchvision shim: transforms.{name} is not available (import-only stand-in)')
    _T.__name__ = name
    return _T


class InterpolationMode:
    NEAREST, BILINEAR, BICUBIC = 'nearest', 'bilinear', 'bicubic'


Normalize, Compose, RandomResizedCrop, ToTensor, Resize, CenterCrop, RandomHorizontalFlip, RandomCrop, ColorJitter, RandomApply, RandomGrayscale = (
    _unavailable(n) for n in ('Normalize', 'Compose', 'RandomResizedCrop', 'ToTensor', 'Resize', 'CenterCrop', 'RandomHorizontalFlip', 'RandomCrop',
                              'ColorJitter', 'RandomApply', 'RandomGrayscale'))
