"""MI355X-native two-stream FCN + probabilistic-fusion engine (hot path of
ethz-asl/modular_semantic_segmentation behind its BaseModel API).  See DESIGN.md."""
__version__ = '0.1.0'


def get_model(name):
    """Model registry with the reference's names (xview/models/__init__.py:10-26)."""
    from .simple_fcn import SimpleFCN
    from .bayes_mix import BayesFusion
    from .dirichlet_mix import DirichletFusion
    from .average_mix import AverageFusion
    from .fusion_fcn import FusionFCN
    from .adapnet import Adapnet
    if name == 'fcn':
        return SimpleFCN
    elif name == 'fusion_fcn':
        return FusionFCN
    elif name == 'adapnet':
        return Adapnet
    elif name in ['bayes_mix', 'bayes_fusion']:
        return BayesFusion
    elif name in ['dirichlet_mix', 'dirichlet_fusion']:
        return DirichletFusion
    elif name in ['average_fusion', 'average_mix']:
        return AverageFusion
    raise UserWarning('ERROR: Model %s not found' % name)
