import numpy as np
import pandas as pd


def create_score_dataframe(scores, timestamps, event_classes):
    scores = np.asarray(scores)
    timestamps = np.asarray(timestamps)
    return pd.DataFrame(np.concatenate((timestamps[:-1, None], timestamps[1:, None], scores), axis=1),
                        columns=["onset", "offset", *event_classes])


def validate_score_dataframe(scores, timestamps=None, event_classes=None):
    return scores
