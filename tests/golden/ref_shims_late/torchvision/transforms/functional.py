def __getattr__(name):
    raise NotImplementedError(f'torchvision shim: transforms.functional.{name} is not available (import-only stand-in)')
