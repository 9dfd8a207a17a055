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
    """Per-episode history of the ``info`` counters and their means: the interface of the reference's handler of the same
    name (src/metrics.py:67-95 -- ``step / set / compute / save_metrics / load_metrics``, JSON of ``{name: [values]}``),
    extended for the batched env: ``step(info, ended)`` takes ``info`` as a dict of ``[B]`` tensors plus the mask of
    envs whose episode just finished and records one entry per finished episode, which is what ``train()`` does at every
    episode end (src/train.py:427-430).  Whole-node totals without histories: ``dist.node_metrics``."""

    def __init__(self):
        self.metrics = {m: [] for m in SusMetrics}

    def _history(self, name):
        key = SusMetrics(name) if not isinstance(name, SusMetrics) else name
        return self.metrics.setdefault(key, [])

    def step(self, metrics, ended=None) -> None:
        for name, value in metrics.items():
            history = self._history(name)
            if hasattr(value, "tolist"):
                picked = value if ended is None else value[ended]
                history.extend(picked.reshape(-1).tolist())
            else:
                history.append(value)

    def set(self, metrics) -> None:
        known = {m.value for m in SusMetrics}
        for name, values in metrics.items():
            assert str(name) in known, f"Invalid metric: {name}"
            self.metrics[SusMetrics(name)] = values

    def compute(self):
        return {name: sum(history) / len(history) for name, history in self.metrics.items()}

    def save_metrics(self, save_file_path) -> None:
        import json
        from pathlib import Path

        Path(save_file_path).write_text(json.dumps({str(name): history for name, history in self.metrics.items()}))

    def load_metrics(self, metrics_file_path) -> None:
        import json
        from pathlib import Path

        self.metrics = json.loads(Path(metrics_file_path).read_text())
