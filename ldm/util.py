from adaface_amd.ldm.util import *  # noqa: F401,F403
from adaface_amd.ldm.util import instantiate_from_config, get_obj_from_str, load_model_from_config, load_config  # noqa: F401
