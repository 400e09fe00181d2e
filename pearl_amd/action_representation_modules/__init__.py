from .action_representation_module import ActionRepresentationModule
from .identity_action_representation_module import IdentityActionRepresentationModule
from .one_hot_action_representation_module import OneHotActionTensorRepresentationModule

__all__ = ["ActionRepresentationModule", "IdentityActionRepresentationModule",
           "OneHotActionTensorRepresentationModule"]
