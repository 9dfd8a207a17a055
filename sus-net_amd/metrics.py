"""Names of the per-episode counters the env returns as ``info`` (reference src/metrics.py:7-32).

``EpisodicMetricHandler`` (reference src/metrics.py:67-95) is trainer bookkeeping and out of scope; its
device analogue is the per-env lifetime accumulators summed by ``susnet_reduce_lifetime``.
"""
from enum import Enum


class SusMetrics(str, Enum):
    IMP_KILLED_CREW = "imp_killed_crew"
    IMP_VOTED_OUT = "imp_voted_out"
    CREW_VOTED_OUT = "crew_voted_out"
    SABOTAGED_JOBS = "sabotaged_jobs"
    COMPLETED_JOBS = "completed_jobs"
    TOTAL_STALEMATES = "total_stalemates"
    TOTAL_TIME_STEPS = "total_time_steps"
    IMPOSTER_WON = "imposter_won"
    CREW_WON = "crew_won"
    AVG_CREW_RETURNS = "avg_crew_returns"
    AVG_IMPOSTER_RETURNS = "avg_imposter_returns"
    CREW_LOSS = "crew_loss"
    IMPOSTER_LOSS = "imposter_loss"

    def __str__(self):
        return self.value

    @classmethod
    def can_increment(cls, metric: str):  # reference src/metrics.py:22-32
        return metric in [cls.IMP_KILLED_CREW, cls.IMP_VOTED_OUT, cls.CREW_VOTED_OUT, cls.SABOTAGED_JOBS,
                          cls.COMPLETED_JOBS, cls.TOTAL_STALEMATES, cls.TOTAL_TIME_STEPS]
