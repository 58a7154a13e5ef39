"""Model discovery by name, as Module2/models/__init__.py:25-67: ``--model X`` imports
``models/X_model.py`` and picks the BaseModel subclass whose lower-cased name is ``X`` without
underscores + ``model``."""
import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    try:
        modellib = importlib.import_module('animateportrait_amd.models.' + model_name + '_model')
    except ImportError as e:
        raise NotImplementedError('model [%s] is not part of the MI355X hot path (%s)' % (model_name, e))
    target = model_name.replace('_', '') + 'model'
    model = None
    for name, cls in modellib.__dict__.items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            model = cls
    if model is None:
        raise NotImplementedError('In %s_model.py, there should be a subclass of BaseModel with class name that '
                                  'matches %s in lowercase.' % (model_name, target))
    return model


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    model = find_model_using_name(opt.model)
    instance = model(opt)
    print('model [%s] was created' % type(instance).__name__)
    return instance
