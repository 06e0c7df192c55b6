"""JSON / class-name object factory — txt2vid/util/reflection.py:12-50. This is the reference's plugin
boundary: `--G --D --sent --data --D_loss --G_loss` name classes (or JSON files {"class", "args"}).
Names under the reference's package (`txt2vid.…`) resolve to this package's drop-in classes."""
import importlib
import json
from pathlib import Path

ALIAS_FROM, ALIAS_TO = 'txt2vid.', 'txt2vid_amd.'


def get_class(kls):
    if kls.startswith(ALIAS_FROM):
        kls = ALIAS_TO + kls[len(ALIAS_FROM):]
    module, _, name = kls.rpartition('.')
    return getattr(importlib.import_module(module), name)


def create_object(json_or_file, **kwargs):
    if isinstance(json_or_file, str):
        if Path(json_or_file).exists():
            return create_object_file(json_or_file, **kwargs)
        return create_object_json({'class': json_or_file}, **kwargs)
    assert isinstance(json_or_file, dict)
    return create_object_json(json_or_file, **kwargs)


def create_object_json(json_obj, **kwargs):
    clz = get_class(json_obj['class'])
    args = dict(json_obj.get('args', {}))
    args.update(kwargs)
    return clz(**args)


def create_object_file(json_file_path, **kwargs):
    with open(json_file_path) as f:
        params = json.load(f)
    assert 'class' in params
    return create_object(params, **kwargs)


def alias_reference_modules():
    """Let pickles written by the reference (whole-module checkpoints such as `--sent_weights`: train/txt.py:185 saves
    `{'optim': optimizer, 'txt': seq2seq}`) resolve their `txt2vid.*` class paths to this package's drop-in classes."""
    import sys
    for ours in ('', '.models', '.models.txt', '.models.txt.basic', '.models.layers', '.models.conv_lstm', '.models.resnet3d',
                 '.models.tganv2', '.models.tganv2.gen', '.models.tganv2.discrim', '.models.tganv2_cond', '.models.tganv2_cond.gen',
                 '.models.tganv2_cond.discrim', '.gan', '.gan.losses', '.gan.cond_gan', '.data', '.util', '.util.pick'):
        theirs = 'txt2vid' + ours
        if theirs not in sys.modules:
            try:
                sys.modules[theirs] = importlib.import_module('txt2vid_amd' + ours)
            except ImportError:
                pass
