"""Planner-side operators of the reference (`planning/astar.py`) that sit either side of view scoring."""
from planning.astar import AstarPlanner, OccupancyOps  # noqa: F401
