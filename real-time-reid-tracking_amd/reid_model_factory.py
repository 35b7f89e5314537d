"""Plugin surface of the yolov8_tracking / StrongSORT re-ID loader - mirror of
modification_tracking/reid_model_factory.py (function names, argument meaning and error behaviour).

Differences that matter:
  * the model-type list additionally knows this engine's native backbone ("seres18_ibn"), so weight files named
    after it resolve; the reference's additions ("vit", "swin_transformer", :9) are kept;
  * the latent NameErrors of the reference's download / warn paths (``time``, ``sys``, ``warnings``
    are never imported there) are not reproduced;
  * ``load_pretrained_weights`` accepts this package's backbone objects as well as ``torch.nn.Module``.
"""
import sys
import time
import warnings
from collections import OrderedDict

# native backbones first: "seres18_ibn" must win over shorter substrings
__model_types = [
    'seres18_ibn',
    'resnet50', 'mlfn', 'hacnn', 'mobilenetv2_x1_0', 'mobilenetv2_x1_4',
    'osnet_x1_0', 'osnet_x0_75', 'osnet_x0_5', 'osnet_x0_25',
    'osnet_ibn_x1_0', 'osnet_ain_x1_0', "vit", "swin_transformer"]

_DRIVE = 'https://drive.google.com/uc?id='
# effective content of the reference's weight-URL table (reid_model_factory.py:11-100, later duplicates win),
# keyed (architecture, dataset) -> Google-Drive file id
_DRIVE_IDS = {
    'resnet50': ('1dUUZ4rHDWohmsQXCRe2C_HbYkzz94iBV', '17ymnLglnc64NRvGOitY3BqMRS9UWd1wg', '1yiBteqgIZoOeywE8AhGmEQl7FTVwrQmf'),
    'resnet50_fc512': ('1kv8l5laX_YCdIGVCetjlNdzKIA3NvsSt', '13QN8Mp3XH81GK4BPGXobKHKyTGH50Rtx', '1fDJLcz4O5wxNSUvImIIjoaIF9u1Rwaud'),
    'mlfn': ('1wXcvhA_b1kpDfrt9s2Pma-MHxtj9pmvS', '1rExgrTNb0VCIcOnXfMsbwSUW1h2L1Bum', '18JzsZlJb3Wm7irCbZbZ07TN4IFKvR6p-'),
    'hacnn': ('1LRKIQduThwGxMDQMiVkTScBwR7WidmYF', '1zNm6tP4ozFUCUQ7Sv1Z98EAJWXJEhtYH', '1MsKRtPM5WJ3_Tk2xC0aGOO7pM3VaFDNZ'),
    'mobilenetv2_x1_0': ('18DgHC2ZJkjekVoqBWszD8_Xiikz-fewp', '1q1WU2FETRJ3BXcpVtfJUuqq4z3psetds', '1j50Hv14NOUAg7ZeB3frzfX-WYLi7SrhZ'),
    'mobilenetv2_x1_4': ('1t6JCqphJG-fwwPVkRLmGGyEBhGOf2GO5', '12uD5FeVqLg9-AFDju2L7SQxjmPb4zpBN', '1ZY5P2Zgm-3RbDpbXM0kIBMPvspeNIbXz'),
    'osnet_x1_0': ('1vduhq5DpN2q1g4fYEZfPI17MJeh9qyrA', '1QZO_4sNf4hdOKKKzKc-TZU9WW1v6zQbq', '1IosIFlLiulGIjwW3H8uMRmx3MzPwf86x'),
    'osnet_x0_75': ('1ozRaDSQw_EQ8_93OUmjDbvLXw9TnfPer', '1IE3KRaTPp4OUa6PGTFL_d5_KQSJbP0Or', '1fhjSS_7SUGCioIf2SWXaRGPqIY9j7-uw'),
    'osnet_x0_5': ('1PLB9rgqrUM7blWrg4QlprCuPT7ILYGKT', '1KoUVqmiST175hnkALg9XuTi1oYpqcyTu', '1DHgmb6XV4fwG3n-CnCM0zdL9nMsZ9_RF'),
    'osnet_x0_25': ('1z1UghYvOTtjx7kEoRfmqSMu-z62J6MAj', '1eumrtiXT4NOspjyEV4j8cHmlOaaCGk5l', '1Kkx2zW89jq_NETu4u42CFZTMVD5Hwm6e'),
    'osnet_ibn_x1_0': (None, None, '1q3Sj2ii34NlfxA4LvmHdWO_75NDRmECJ'),
    'osnet_ain_x1_0': (None, None, '1SigwBE6mPdqiJMqhuIY4aqC7--5CsMal'),
}
__trained_urls = OrderedDict()
for _arch, _ids in _DRIVE_IDS.items():
    for _ds, _id in zip(('market1501', 'dukemtmcreid', 'msmt17'), _ids):
        if _id is not None:
            __trained_urls['%s_%s.pt' % (_arch, _ds)] = _DRIVE + _id


def show_downloadeable_models():
    print('\nAvailable .pt ReID models for automatic download')
    print(list(__trained_urls.keys()))


def get_model_url(model):
    """URL for a known weight file name (``model`` is a pathlib.Path), else None."""
    return __trained_urls.get(model.name)


def is_model_in_model_types(model):
    return model.name in __model_types


def get_model_name(model):
    """First model type that is a substring of the file name (reid_model_factory.py:122-126), else None."""
    for x in __model_types:
        if x in model.name:
            return x
    return None


def download_url(url, dst):
    """Downloads ``url`` to ``dst`` with a progress line (reid_model_factory.py:129-155)."""
    from urllib import request
    print('* url="{}"'.format(url))
    print('* destination="{}"'.format(dst))
    state = {}

    def _reporthook(count, block_size, total_size):
        if count == 0:
            state['t0'] = time.time()
            return
        duration = max(time.time() - state['t0'], 1e-9)
        done = int(count * block_size)
        sys.stdout.write('\r...%d%%, %d MB, %d KB/s, %d seconds passed' % (
            int(done * 100 / max(total_size, 1)), done / (1024 * 1024), int(done / (1024 * duration)), duration))
        sys.stdout.flush()

    request.urlretrieve(url, dst, _reporthook)
    sys.stdout.write('\n')


def load_pretrained_weights(model, weight_path):
    """Loads pretrained weights into ``model``; never raises on a mismatch (reid_model_factory.py:158-210):
      * accepts ``{'state_dict': ...}`` or a bare dict (:172-175), strips ``module.`` (:182-183);
      * copies only entries whose name AND shape match (:185-189);
      * warns when nothing matched (:194-199), otherwise prints the discarded keys (:205-210).
    """
    import torch
    checkpoint = torch.load(weight_path, map_location='cpu')
    state_dict = checkpoint['state_dict'] if 'state_dict' in checkpoint else checkpoint
    model_dict = model.state_dict()
    new_state_dict = OrderedDict()
    matched_layers, discarded_layers = [], []
    for k, v in state_dict.items():
        if k.startswith('module.'):
            k = k[7:]
        if k in model_dict and tuple(model_dict[k].shape) == tuple(v.shape):
            new_state_dict[k] = v
            matched_layers.append(k)
        else:
            discarded_layers.append(k)
    model_dict.update(new_state_dict)
    model.load_state_dict(model_dict)
    if len(matched_layers) == 0:
        warnings.warn('The pretrained weights "{}" cannot be loaded, please check the key names manually '
                      '(** ignored and continue **)'.format(weight_path))
    else:
        print('Successfully loaded pretrained weights from "{}"'.format(weight_path))
        if len(discarded_layers) > 0:
            print('** The following layers are discarded due to unmatched keys or layer size: {}'.format(discarded_layers))
