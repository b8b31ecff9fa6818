"""Model registry with the reference's surface (bpmult/models/__init__.py:6-14)."""
from .bpmult import MultiprojectionMMTransformer3DGMUClf, MultiprojectionMMTransformerGMUClf
from .encoder import TransformerEncoder

MODELS = {
    "mmtrvapt": MultiprojectionMMTransformerGMUClf,
    "mmtrvat": MultiprojectionMMTransformer3DGMUClf,
}


def get_model(args, config=None):
    return MODELS[args.model](args)
