"""shasta_amd: MI355X-native implementation of ShaSTA's learned track<->detection affinity path.

Host code mirrors the reference's model/registry API (tools/nusc_shasta call surface); compute is hand-written HIP
for gfx950 behind the C ABI in include/shasta_hip.h.  See DESIGN.md and INTEGRATION.md."""
from . import hip  # noqa: F401
from .bird_eye_view import BEVFeatureExtractor  # noqa: F401
from .builder import build_simp_track, build_track  # noqa: F401
from .registry import BACKBONES, NECKS, READERS, SECOND_STAGE, TRACK, Registry, build_from_cfg  # noqa: F401
from .shasta import Shasta, load_state_dict_permissive  # noqa: F401
from .train_track import example_to_device, track_batch_processor  # noqa: F401
from .voxel_encoder import VoxelFeatureExtractorV3  # noqa: F401

__version__ = "0.1.0"
