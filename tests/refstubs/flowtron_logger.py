"""Stand-in for the reference's flowtron_logger.py (tensorboard + matplotlib are not installed here; logging is outside the
hot path).  train.py imports the name unconditionally (train.py:28) and only instantiates it when
`with_tensorboard` is set."""


class FlowtronLogger:
    def __init__(self, logdir):
        self.logdir = logdir
        self.scalars = []

    def add_scalar(self, tag, value, step):
        self.scalars.append((tag, float(value), int(step)))

    def log_training(self, loss, learning_rate, iteration):
        self.add_scalar("training/loss", loss, iteration)

    def log_validation(self, *args, **kwargs):
        pass
