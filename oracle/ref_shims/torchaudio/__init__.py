from . import compliance  # noqa: F401
from . import transforms  # noqa: F401
