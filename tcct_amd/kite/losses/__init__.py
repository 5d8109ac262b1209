from .loss import DiceLoss, MultiLoss, get_loss      # noqa: F401
from .miou import MDiceLoss, MIouLoss, MaskOneHot    # noqa: F401
