from .all_reduce import AllReduce, all_reduce_sum_

__all__ = ["AllReduce", "all_reduce_sum_"]
