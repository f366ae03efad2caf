"""The reference's experiment graphs as ``Model`` subclasses.

Each class carries what its reference file's ``ExperimentModel._build_graph`` and ``__main__`` fix:
encoder layers, number of ConvLSTM layers, history length, sequence length, head width and the
decode / IoU variant.  ``make(name, ...)`` builds one with the reference's hyper-parameters.

 name               reference file (polyrnn/)                  encoder                hist  T   L   head
 History0           History0Experiment.py:20-57,181-184        16@5,16@5,16@3          0   10   2   P^2
 History1Large      History1LargeExperiment.py:13-50,70-73     16@5,16@5,16@3          1   10   2   P^2
 History2Large      History2LargeExperiment.py:13-50,70-73     16@5,16@5,16@3          2   10   2   P^2
 RNNConv            RNNConvExperiment.py:12-48,70-73           32@14,32@5,32@3         1    5   2   P^2
 DeepRNN            DeepRNNExperiment.py:13-50,70-76           16@5,16@5,16@3          2   10  10   P^2
 EndToken           EndTokenExperiment.py:20-58,200-203        16@5,16@5,16@3          2   10   2   P^2+1
 EndTokenDeep_2     EndTokenDeepExperiment_2.py:19-57,202-205  16@5,16@5,5x16@3        2   10   5   P^2+1
"""
from .Model import Model


class History2Large(Model):
    encoder_spec = ((16, 5), (16, 5), (16, 3))
    rnn_layers = 2
    defaults = dict(max_timesteps=10, history_length=2)


class History1Large(History2Large):
    defaults = dict(max_timesteps=10, history_length=1)


class History0(History2Large):
    """Own validate_iou: the predicted polygon is the predictions alone
    (reference polyrnn/History0Experiment.py:158)."""
    defaults = dict(max_timesteps=10, history_length=0)
    iou_prepend_history = False


class RNNConv(Model):
    encoder_spec = ((32, 14), (32, 5), (32, 3))
    rnn_layers = 2
    defaults = dict(max_timesteps=5, history_length=1)


class DeepRNN(History2Large):
    rnn_layers = 10


class EndToken(History2Large):
    """P*P+1 classes; decoding stops on the end token (P, 0), which is replaced by the closing
    vertex in the predicted polygon and by point (0, 0) in the next history frame
    (reference polyrnn/EndTokenExperiment.py:158-170)."""
    end_token = True


class EndTokenDeep_2(EndToken):
    """Deeper encoder and recurrence; the IoU polygon keeps only the first ``duration`` predictions
    (reference polyrnn/EndTokenDeepExperiment_2.py:174-177)."""
    encoder_spec = ((16, 5), (16, 5), (16, 3), (16, 3), (16, 3), (16, 3), (16, 3))
    rnn_layers = 5
    iou_slice_by_duration = True


EXPERIMENTS = {c.__name__: c for c in (History0, History1Large, History2Large, RNNConv, DeepRNN, EndToken,
                                       EndTokenDeep_2)}


def make(name, image_size=28, filepath=None, sess=None, **kwargs):
    """Builds experiment ``name`` with the reference's max_timesteps / history_length."""
    cls = EXPERIMENTS[name]
    d = dict(cls.defaults)
    d.update({k: kwargs.pop(k) for k in ("max_timesteps", "history_length") if k in kwargs})
    return cls(sess, d["max_timesteps"], image_size, image_size, d["history_length"], filepath, **kwargs)
