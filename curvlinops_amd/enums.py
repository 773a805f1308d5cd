"""String enums selecting the Fisher flavour and the KFAC weight-sharing approximation
(reference ``curvlinops/kfac_utils.py:20-75``); ``"mc" in FisherType`` works like there."""

from enum import Enum, EnumMeta


class _Contains(EnumMeta):
    def __contains__(cls, item) -> bool:
        try:
            cls(item)
        except ValueError:
            return False
        return True


class FisherType(str, Enum, metaclass=_Contains):
    TYPE2 = "type-2"
    MC = "mc"
    EMPIRICAL = "empirical"
    FORWARD_ONLY = "forward-only"


class KFACType(str, Enum, metaclass=_Contains):
    EXPAND = "expand"
    REDUCE = "reduce"
