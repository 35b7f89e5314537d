"""torchreid-style model registry - mirror of modification_tracking/models/__init__.py:26-121.

The reference's registry maps ~55 names to torchreid constructors that are not vendored; its
own additions are 'vit' and 'swin_transformer' (:79-80).  This registry holds the backbones that exist as HIP
kernel sequences; every other name raises the same ``KeyError`` the reference raises for an unknown model.
"""
from .backbone import cares18_ibn, emares18_ibn, seres18_ibn, swin_t

__model_factory = {
    'seres18_ibn': seres18_ibn,
    'cares18_ibn': cares18_ibn,          # sibling backbones of reid/backbones (CARes18.py, EMA_Res18.py)
    'emares18_ibn': emares18_ibn,
    'swin_transformer': swin_t,          # the reference's own addition (models/__init__.py:80)
}


def show_avai_models():
    print(list(__model_factory.keys()))


def build_model(name, num_classes, loss='softmax', pretrained=True, use_gpu=True, precision=None):
    """Same signature, same ``KeyError`` text as models/__init__.py:93-121; constructors are called with
    (num_classes, loss, pretrained, use_gpu).  ``precision`` (keyword, this engine's addition): the arithmetic the model runs
    in - None = $REID_PRECISION, else "f16x3", the mode bench.py reports (precision.py)."""
    avai_models = list(__model_factory.keys())
    if name not in avai_models:
        raise KeyError('Unknown model: {}. Must be one of {}'.format(name, avai_models))
    return __model_factory[name](num_classes=num_classes, loss=loss, pretrained=pretrained, use_gpu=use_gpu, precision=precision)
