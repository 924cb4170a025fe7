from .evaluator import (BitsEvaluator, CodesExtractor, DatasetEvaluator, DatasetEvaluators, MSEEvaluator, VTSampler,
                        build_evaluator, inference_on_dataset)

__all__ = ["BitsEvaluator", "CodesExtractor", "DatasetEvaluator", "DatasetEvaluators", "MSEEvaluator", "VTSampler",
           "build_evaluator", "inference_on_dataset"]
