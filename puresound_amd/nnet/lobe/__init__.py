"""Building blocks of the maskers: parameter holders under the reference's state_dict keys plus their HIP drivers."""
