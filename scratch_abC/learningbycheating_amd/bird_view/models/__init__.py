from .image import ImagePolicyModelSS          # noqa: F401
from .birdview import BirdViewPolicyModelSS    # noqa: F401
