from .vgg import VGG, vgg16  # noqa: F401
