"""streammind/model/multimodal_projector/builder.py: module path of the projector classes.  `Video_Mamba_seq` (the StreamMind
connector + gate, :390-564) is ../stream_model.py over the native model; the STC family and `build_vision_projector`
(:119-158, 574-796) are ../stc_connector.py."""
from ..stc_connector import (MlpGeluProjector, SpatialConv, SpatialPool, STCConnector, STCConnectorV35, STPConnector,  # noqa: F401
                             build_vision_projector)
from ..stream_model import Video_Mamba_seq  # noqa: F401
