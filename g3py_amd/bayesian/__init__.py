from .selection import Experiment, random_obs, uniform_obs
