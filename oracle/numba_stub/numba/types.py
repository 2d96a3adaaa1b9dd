"""Type placeholders used only as arguments to ``typed.List.empty_list``."""
intp = int
int64 = int
float64 = float
