"""Event record layout shared by the host mirror, the synthetic generators and the tests.

Mirrors the in-memory layout of ``dvs_msgs::Event``
(reference: feature_tracker/src/dvs_msgs/Event.h:42-52): ``uint16 x; uint16 y;
ros::Time ts {uint32 sec; uint32 nsec}; uint8 polarity`` -> 16 bytes, AoS.
"""
import numpy as np

EVENT_DTYPE = np.dtype(
    {
        "names": ["x", "y", "sec", "nsec", "polarity"],
        "formats": ["<u2", "<u2", "<u4", "<u4", "u1"],
        "offsets": [0, 2, 4, 8, 12],
        "itemsize": 16,
    }
)


def make_events(x, y, t_us, polarity):
    """Pack arrays into an EVENT_DTYPE array. ``t_us`` are integer microseconds."""
    x = np.asarray(x)
    n = x.shape[0]
    ev = np.zeros(n, dtype=EVENT_DTYPE)
    t_us = np.asarray(t_us, dtype=np.int64)
    ev["x"] = x
    ev["y"] = np.asarray(y)
    ev["sec"] = (t_us // 1_000_000).astype(np.uint32)
    ev["nsec"] = ((t_us % 1_000_000) * 1000).astype(np.uint32)
    ev["polarity"] = np.asarray(polarity).astype(np.uint8)
    return ev


def event_times(ev):
    """ros::Time::toSec(): (double)sec + 1e-9*(double)nsec (two roundings, no FMA)."""
    return ev["sec"].astype(np.float64) + 1e-9 * ev["nsec"].astype(np.float64)
