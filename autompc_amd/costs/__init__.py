from .cost import Cost, QuadCost
from .sum_cost import SumCost
from .thresh_cost import ThresholdCost, BoxThresholdCost
from .terms import cost_terms

__all__ = ["Cost", "QuadCost", "SumCost", "ThresholdCost", "BoxThresholdCost", "cost_terms"]
