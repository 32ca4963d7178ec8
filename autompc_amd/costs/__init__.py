from .cost import Cost, QuadCost
from .sum_cost import SumCost
from .thresh_cost import ThresholdCost, BoxThresholdCost

__all__ = ["Cost", "QuadCost", "SumCost", "ThresholdCost", "BoxThresholdCost"]
