"""create_model / create_loss — the reference's registry contracts (opv2v/opencood/tools/train_utils.py:102-135 and
:138-170): hypes['model' | 'loss']['core_method'] names a module, the class whose lower-cased name equals the name without
underscores is instantiated with hypes[...]['args']."""
import importlib


def _import_or_none(dotted):
    """The module, or None when exactly that module does not exist; an import error raised INSIDE an existing module
    (a missing dependency, a broken relative import) propagates with its real cause."""
    try:
        return importlib.import_module(dotted)
    except ModuleNotFoundError as e:
        if e.name != dotted:
            raise
        return None


def _lookup(package, module_name, what):
    lib = _import_or_none(package + "." + module_name)
    target = module_name.replace("_", "").lower()
    if lib is not None:
        for name, cls in lib.__dict__.items():
            if name.lower() == target and isinstance(cls, type):
                return cls
    # the reference prints and exit(0)s; a missing class is an error here
    raise ValueError("%s not found in %s: a module named %s with a class named %s (ignoring case) is required"
                     % (what, package, module_name, target))


def create_loss(hypes):
    """hypes['loss'] = {core_method: 'vanilla_seg_loss', args: {...}} -> the forward-only loss mirror (validation loss)"""
    return _lookup("cobevt_amd.host", hypes["loss"]["core_method"], "loss function")(hypes["loss"]["args"])


def create_model(hypes):
    backbone_name = hypes["model"]["core_method"]
    backbone_config = hypes["model"]["args"]
    model_lib = _import_or_none("cobevt_amd.host." + backbone_name)
    model = None
    target_model_name = backbone_name.replace("_", "")
    if model_lib is not None:
        for name, cls in model_lib.__dict__.items():
            if name.lower() == target_model_name.lower() and isinstance(cls, type):
                model = cls
    if model is None:
        # the reference prints and exit(0)s; a missing model is an error here
        raise ValueError("backbone not found in cobevt_amd.host: a module named %s with a class named %s is "
                         "required (FAX hot path models: corpbevt, fax_fused_transformer)"
                         % (backbone_name, target_model_name))
    return model(backbone_config)
