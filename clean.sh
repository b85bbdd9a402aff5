#!/bin/sh
# clean.sh — deletes the database directory recorded in ./.DBDirectory and the local settings (CuCLARK's clean.sh:24-41).
if [ ! -s ./.DBDirectory ]; then
  echo "There is no database directory to clean"
  exit 0
fi
DIR=$(head -n 1 ./.DBDirectory)
echo "Are you sure you want to delete all data in the database directoy: $DIR? (yes/no)"
read -r decision
case "$decision" in
  yes|y|Y|Yes|YES)
    echo "Cleaning: on-going..."
    if [ -n "$DIR" ] && [ -d "$DIR" ] && [ "$(cd "$DIR" && pwd)" != "/" ]; then
      find "$DIR" -delete
    fi
    for f in .dbAddress .DBDirectory .settings; do
      [ -f "$f" ] && unlink "$f"
    done
    echo "Cleaning: done." ;;
  no|n|N|No|NO) echo "Cleaning: canceled" ;;
esac
