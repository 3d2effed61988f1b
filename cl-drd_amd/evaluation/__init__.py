from .retrieval_evaluator import RankingEvaluator  # noqa: F401
