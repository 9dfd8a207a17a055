"""Names of the per-episode counters the env returns as ``info`` (reference src/metrics.py:7-32).

``EpisodicMetricHandler`` (reference src/metrics.py:67-95, trainer bookkeeping) is mirrored below for drop-in use; the
device-side analogue for whole-node totals is the per-env lifetime accumulators summed by ``susnet_reduce_lifetime``.
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


class EpisodicMetricHandler:
    """Averages the ``info`` counters over finished episodes (reference src/metrics.py:67-95, same methods and JSON
    format).  ``step`` also takes the batched env's ``info`` (a dict of ``[B]`` tensors) together with the mask of envs
    whose episode just ended: one entry per finished episode is appended, as ``train()`` does at every episode end
    (src/train.py:427-430).  For whole-node totals without per-episode lists see ``dist.node_metrics``."""

    def __init__(self):
        self.metrics = {metric: [] for metric in SusMetrics}

    def step(self, metrics, ended=None) -> None:
        for metric, value in metrics.items():
            if hasattr(value, "tolist"):  # [B] tensor / array: the episodes that ended in this tick
                value = value[ended] if ended is not None else value
                self.metrics[metric].extend(value.reshape(-1).tolist())
            else:
                self.metrics[metric].append(value)

    def set(self, metrics) -> None:
        for metric, values in metrics.items():
            assert any(m.value == metric for m in SusMetrics), f"Invalid metric: {metric}"
            self.metrics[metric] = values

    def compute(self):
        return {metric: sum(values) / len(values) for metric, values in self.metrics.items()}

    def save_metrics(self, save_file_path):
        import json

        with open(save_file_path, "w") as f:
            json.dump({str(k): v for k, v in self.metrics.items()}, f)

    def load_metrics(self, metrics_file_path):
        import json

        with open(metrics_file_path, "r") as f:
            self.metrics = json.load(f)
