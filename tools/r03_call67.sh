bash tools/r03_call6.sh; bash tools/r03_call7.sh
