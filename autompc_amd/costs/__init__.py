from .cost import Cost, QuadCost
from .sum_cost import SumCost
from .thresh_cost import ThresholdCost, BoxThresholdCost
from .terms import cost_terms
from .blocks import quad_sum_block, is_quad_sum

__all__ = ["Cost", "QuadCost", "SumCost", "ThresholdCost", "BoxThresholdCost", "cost_terms",
           "quad_sum_block", "is_quad_sum"]
