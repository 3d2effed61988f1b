from .nway_listwise import NwayTrainer, linear_schedule_factor, no_decay  # noqa: F401
